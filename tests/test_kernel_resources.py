"""Compile-time guard for the two product row kernels (CPU, no GPU needed): hipcc's
kernel-resource-usage remarks must show the register/occupancy shape the launch code and the
measurements in HISTORY.md section 5 assume.  The f64 kernel sits exactly at the 256-VGPR limit,
where an innocent source change can flip the allocator into tens of spills (a wave-uniform store
offset moved to the SGPR operand did: 0 -> 50 spills)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "caf_cookoff_amd" / "csrc"


def _usage():
    if not (shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists()):
        pytest.skip("hipcc not available")
    subprocess.run(["make", "-C", str(CSRC), "asm"], check=True, capture_output=True, timeout=900)
    text = (CSRC / "build" / "resource_usage.txt").read_text()
    out = {}
    for m in re.finditer(r"Function Name: (\S+)(.*?)(?=Function Name:|\Z)", text, re.S):
        fields = dict(re.findall(r"remark:\s+([A-Za-z \[\]/]+?): (\S+) \[-Rpass", m.group(2)))
        out[m.group(1)] = fields
    return out


def test_product_row_kernels_have_no_vgpr_spills_and_expected_occupancy():
    usage = _usage()
    f64 = usage["_ZN3caf10k_seq_rowsIdLi15ENS_5SeqIoIdEEEEvNS_9FusedArgsIT_EEPKNS_3cpxIS4_EE"]
    f32 = usage["_ZN3caf10k_duo_rowsIfNS_5DuoIoIfEEEEvNS_9FusedArgsIT_EEPKNS_3cpxIS4_EE"]
    assert int(f64["VGPRs Spill"]) == 0 and int(f64["Occupancy [waves/SIMD]"]) == 2
    assert int(f64["LDS Size [bytes/block]"]) * 2 <= 160 * 1024       # two workgroups per CU
    assert int(f32["VGPRs Spill"]) == 0 and int(f32["Occupancy [waves/SIMD]"]) == 3
    assert int(f32["LDS Size [bytes/block]"]) * 3 <= 160 * 1024       # three workgroups per CU
    # the haystack-spectrum kernels share the LDS geometry
    for name, fields in usage.items():
        if "k_seq_prepare" in name:
            assert int(fields["VGPRs Spill"]) == 0


def test_chain_kernels_register_shape():
    """The size-generic chain kernels (kernels_chain.hpp): LDS per workgroup within 160 KiB, the
    occupancy the launch code assumes (complex128: 2 waves per SIMD = 256 VGPRs, complex64: 4 = 128
    VGPRs), and spills bounded -- the R = 2 instantiations run (almost) spill-free since the butterfly
    constants became SGPR operands; the R = 4 complex64 kernel (BASELINE configs[3]) sits at
    35 spilled registers (90 before the load/store optimizer was switched off for this kernel family) and
    the R = 8 kernels (six live slab descriptors + the W_128 combine; chain pairs as a run-time loop) at 59 / 74, known costs
    (HISTORY.md section 5) that must not get worse silently."""
    usage = _usage()
    seen = 0
    for name, f in usage.items():
        if "k_chain_rows" not in name:
            continue
        seen += 1
        is_f64 = "k_chain_rowsId" in name
        r = 16 if "ELi16ELi1ELi0E" in name else 8 if "ELi8ELi1ELi0E" in name else 4 if "ELi4ELi1ELi0E" in name else 2
        assert int(f["LDS Size [bytes/block]"]) <= 160 * 1024
        assert int(f["Occupancy [waves/SIMD]"]) == (2 if is_f64 else 4), name
        limit = {2: 30, 4: 30 if is_f64 else 50, 8: 70 if is_f64 else 90, 16: 10 if is_f64 else 60}[r]
        assert int(f["VGPRs Spill"]) <= limit, (name, f["VGPRs Spill"])
    assert seen == 13  # 6 complex128 + 7 complex64 instantiations (R = 16: n = 65536 complex128, n = 131072 complex64)


def test_small_row_kernels_register_shape():
    """k_small_rows (kernels_small.hpp, n = 8 ... 512): the launch code assumes 512-thread workgroups with
    complex128 at two waves per SIMD (one workgroup of <= 160 KiB of LDS per CU) and complex64 at four (two
    workgroups per CU: <= 128 VGPRs, <= 80 KiB each); spills stay marginal."""
    usage = _usage()
    seen = 0
    for name, f in usage.items():
        if "k_small_rows" not in name:
            continue
        seen += 1
        is_f64 = "k_small_rowsId" in name
        lds = int(f["LDS Size [bytes/block]"])
        assert int(f["Occupancy [waves/SIMD]"]) == (2 if is_f64 else 4), name
        assert lds <= (160 if is_f64 else 80) * 1024, (name, lds)
        assert int(f["VGPRs Spill"]) <= (0 if is_f64 else 8), (name, f["VGPRs Spill"])
    assert seen == 14  # LOGL = 4 ... 10, both dtypes
