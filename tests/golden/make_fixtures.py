#!/usr/bin/env python3
"""Regenerate tests/golden/data/*.c64 by RUNNING the reference's own generator.

The reference writes to the relative path ../data (utils/generate.py:43,51-52)
and its tree is read-only, so it is executed with cwd = tests/golden/_work.
Only its OUTPUT files (data) are kept; no reference source is copied.
Works only where /root/reference exists (the build container, not the GPU box).
"""
import hashlib
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/utils/generate.py")

# sha256 prefixes recorded in SURVEY.md section 8c (numpy 2.2.6 / scipy 1.15.3)
EXPECT = {
    "chirp_0_raw.c64": "935fc1eaddffc517",
    "chirp_0_T+202samp_F+69.25Hz.c64": "88a1566cc07057c5",
    "chirp_4_raw.c64": "728584d790a75a9a",
    "chirp_4_T+70samp_F+82.89Hz.c64": "31a8a844f9ab7635",
    "chirp_9_raw.c64": "076ee9301d7771ce",
    "chirp_9_T+176samp_F+61.49Hz.c64": "b390daf1755a84df",
}


def main():
    if not REF.exists():
        sys.exit("reference generator not present; fixtures are committed, nothing to do")
    work = HERE / "_work"
    work.mkdir(exist_ok=True)
    subprocess.run([sys.executable, str(REF)], cwd=work, check=True)
    shutil.rmtree(work)
    for name, pre in EXPECT.items():
        got = hashlib.sha256((HERE / "data" / name).read_bytes()).hexdigest()[:16]
        print(name, got, "ok" if got == pre else f"MISMATCH (expected {pre})")


if __name__ == "__main__":
    main()
