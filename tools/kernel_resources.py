#!/usr/bin/env python3
"""Print hipcc's kernel-resource-usage remarks (make -C caf_cookoff_amd/csrc asm) per kernel:
VGPRs, spills, scratch, occupancy, LDS.  usage: kernel_resources.py [substring] [measure]
("measure": the measurement build, after make -C caf_cookoff_amd/csrc asm-measure)"""
import re
import subprocess
import sys
from pathlib import Path

which = "resource_usage_measure.txt" if "measure" in sys.argv[2:] else "resource_usage.txt"
txt = (Path(__file__).resolve().parent.parent / "caf_cookoff_amd/csrc/build" / which).read_text()
want = sys.argv[1] if len(sys.argv) > 1 else ""
for m in re.finditer(r"Function Name: (\S+)(.*?)(?=Function Name:|\Z)", txt, re.S):
    f = dict(re.findall(r"remark:\s+([A-Za-z \[\]/]+?): (\S+) \[-Rpass", m.group(2)))
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    if want in name:
        print(f"{name:48s} VGPR {f.get('VGPRs'):>4} AGPR {f.get('AGPRs'):>3} spill {f.get('VGPRs Spill'):>4} "
              f"scratch {f.get('ScratchSize [bytes/lane]'):>5} occ {f.get('Occupancy [waves/SIMD]')} "
              f"LDS {f.get('LDS Size [bytes/block]'):>7} SGPR {f.get('SGPRs')}")
