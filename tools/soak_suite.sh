#!/bin/bash
# tools/soak_suite.sh RUNS OUTDIR [FIRST_INDEX] [pageable]
# The abort hunt of HISTORY.md section 10: RUNS consecutive full `pytest -m gpu` runs, ONE process at a time, uncaptured
# (--capture=no: whatever the HIP / HSA runtime prints to fd 2 before an abort() lands in the log instead of in pytest's
# capture buffer) with AMD_LOG_LEVEL=1 (runtime errors only), the whole log kept per run.  tests/conftest.py installs the
# native-stack SIGABRT handler (tests/cpp/abort_trace.so).  A run that hits its time limit ends the soak: no further GPU
# step is started after a timeout.  summary.txt: one line per run (exit status, pytest's last line).
# Fourth argument `pageable`: the pageable lane -- CAF_TESTS_PAGEABLE_COPIES=1 makes the tests' `pinned_copies` fixture a no-op, so
# torch's own pin-on-the-fly path for Tensor.cpu() / .cuda() is exercised again (the path the round-4 abort sat in), with
# ONE mitigation left active: registered host ranges are whole pages of mmap memory (API rule since ABI 4).
set -u
RUNS=${1:-15}
OUT=${2:-gpurun_out/soak}
FIRST=${3:-1}
LANE=${4:-pinned}
if [ "$LANE" = pageable ]; then export CAF_TESTS_PAGEABLE_COPIES=1; fi
mkdir -p "$OUT"
pass=0; fail=0
for i in $(seq "$FIRST" $((FIRST + RUNS - 1))); do
    log="$OUT/run_$(printf %02d "$i").log"
    AMD_LOG_LEVEL=1 timeout -k 10 420 python -m pytest tests -m gpu -q -p no:cacheprovider --capture=no > "$log" 2>&1
    rc=$?
    last=$(grep -E "passed|failed|error|Aborted|Fatal" "$log" | tail -1)
    echo "run $i ($LANE lane) rc=$rc :: $last" | tee -a "$OUT/summary.txt"
    if [ $rc -eq 0 ]; then pass=$((pass + 1)); else fail=$((fail + 1)); fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "run $i hit its time limit: soak stopped" | tee -a "$OUT/summary.txt"; break; fi
done
echo "soak ($LANE lane): $pass clean, $fail not clean (of $RUNS)" | tee -a "$OUT/summary.txt"
