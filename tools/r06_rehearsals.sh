#!/bin/bash
# Round-6 rehearsals of the N > 1 paths on a ONE-GPU box (functional legs, never scaling figures); from the repo root on the GPU box.
#   eight workers of the one-process path on one GPU (devices 0 x 8: thread fan-out, 50-row shards of 2048 surfaces, host join)
#   four ranks of the torchrun path sharing the one GPU over gloo (the process guard of the pool allows six GPU processes)
set -o pipefail
O=gpurun_out/r06f; mkdir -p $O
timeout -k 10 300 python3 bench.py --gpus 8 --in-process --in-process-devices 0,0,0,0,0,0,0,0 --steps 10 --blocks 1 --cpu-seconds 0.5 > $O/bench_inproc_eight_workers_one_gpu.json 2> $O/bench_inproc_eight.stderr && cp bench_detail.json $O/bench_inproc_eight_workers_one_gpu_detail.json && \
CAF_BENCH_REHEARSE_ON_ONE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 4 --steps 10 --blocks 1 --cpu-seconds 0.5 > $O/bench_n4_rehearsal.json 2> $O/bench_n4_rehearsal.stderr && cp bench_detail.json $O/bench_n4_rehearsal_detail.json && \
wc -c $O/*.json
