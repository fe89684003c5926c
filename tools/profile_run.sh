#!/bin/bash
# Profile recipe for profiles/<name>/ (run on the GPU box through gpurun, from the repo root):
#   tools/profile_run.sh gpurun_out/prof_v5 [extra bench.py flags]
# One rocprofv3 pass per counter group, the program directly after `--`, counters never combined
# with tracing domains other than --kernel-trace.  Pack afterwards with tools/profile_pack.py.
# pmc_ea_* / pmc_l2 (round 5): the L2's memory-side request counters by size and by destination, and the L2 hit rate --
# what rocprofv3 exposes on gfx950 towards an HBM-vs-Infinity-Cache split (it lists no MALL or UMC counter).
set -o pipefail
O=${1:?output dir}; shift
EXTRA=("$@")
# A profiled bench.py must not start other GPU programs: rocprofv3's preload initialises the GPU before bench.py runs, and
# `--gpus N > 1` would start torchrun from that process.  Profile a rank's launch shape with --emulate-rank-of N, or the
# one-process multi-device path with --in-process.
gpus=1; safe=0
for ((i = 0; i < ${#EXTRA[@]}; i++)); do
    case "${EXTRA[i]}" in
        --gpus) gpus=${EXTRA[i+1]:-1} ;;
        --gpus=*) gpus=${EXTRA[i]#--gpus=} ;;
        --in-process|--emulate-rank-of|--emulate-rank-of=*) safe=1 ;;
    esac
done
if [ "$gpus" -gt 1 ] && [ "$safe" -ne 1 ]; then
    echo "profile_run.sh: refusing --gpus $gpus under rocprofv3 without --in-process or --emulate-rank-of (see the comment above)" >&2
    exit 4
fi
B="$PWD/bench.py"
mkdir -p "$O"
export TMPDIR=/tmp
python3 "$B" --no-extra "${EXTRA[@]}" > "$O/bench_default.json" 2>/dev/null || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$B" --steps 20 --warmup 3 --blocks 0 --no-cpu-baseline --no-extra --no-ceiling "${EXTRA[@]}" > "$O/stats_bench.json" 2>/dev/null || exit 2
for grp in "pmc_fetch FETCH_SIZE" "pmc_write WRITE_SIZE" \
           "pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "pmc_sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "pmc_ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum" \
           "pmc_ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum" \
           "pmc_l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    set -- $grp
    d=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$O/$d" -- python3 "$B" --steps 5 --warmup 2 --blocks 0 --no-cpu-baseline --no-extra --no-ceiling "${EXTRA[@]}" > /dev/null 2>&1 || exit 3
    echo "done $d"
done
