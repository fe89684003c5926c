#!/bin/bash
# tools/traffic_components.sh OUTDIR   (run on the GPU box through gpurun, from the repo root)
# Which buffers does the fabric traffic of the configs[3] row kernel (k_chain_rows<float, 14, 4>, 4096 x 65536 complex64) consist
# of?  FETCH_SIZE and WRITE_SIZE of the SAME launch with one kind of global access cut at a time (the measurement library's
# CAF_CHAIN_ABL masks: wrong results, counters and timing only): 0 nothing cut, 4 no slab round trip, 16 no surface stores,
# 2 no haystack-spectrum loads, 8 no needle loads, 30 no global memory at all.  The difference to mask 0 is that buffer's
# share of the bytes that leave L2.  One rocprofv3 pass per counter, the program directly after `--`.
# Summarise with tools/traffic_components.py OUTDIR profiles/<name>/.
set -o pipefail
O=${1:?output dir}
mkdir -p "$O"
export TMPDIR=/tmp
for abl in 0 4 16 2 8 30; do
    for c in FETCH_SIZE WRITE_SIZE; do
        CAF_CHAIN_ABL=$abl rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/abl${abl}_$c" -- python3 tools/chain_time.py 32768 4096 c64 1 > "$O/abl${abl}_$c.txt" 2>&1 || exit 3
    done
    echo "done mask $abl"
done
