#!/bin/bash
# Round 6: what the per-step global-peak exchange of the one-process-per-GPU path costs -- ONE rank under real RCCL on one GPU
# (CAF_BENCH_FORCE_COLLECTIVES=1 under torchrun: process group, barriers and both all-reduces of the N > 1 path execute), the
# library's three element kernels around the collectives (--peak-reduce fused) against ~20 torch tensor operations (allreduce),
# A/B/A/B, and the same launch without any collective.  From the repo root on the GPU box.
O=gpurun_out/r06m; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -k "peak_exchange or rccl_single_rank" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
show='import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "surfaces/s; exchange:", d["config"]["peak_exchange"], "; blocks ms/step:", d["extra"]["headline_blocks"]["ms_per_step_min"], d["extra"]["headline_blocks"]["ms_per_step_median"], d["extra"]["headline_blocks"]["ms_per_step_max"])'
i=0
for m in fused allreduce fused allreduce; do
    i=$((i+1))
    CAF_BENCH_FORCE_COLLECTIVES=1 timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2974$i bench.py --gpus 1 --steps 50 --blocks 3 --no-extra --no-cpu-baseline --peak-reduce $m 2>/dev/null > $O/rccl1_${m}_$i.json
    python3 -c "$show" $O/rccl1_${m}_$i.json "$m"
done
python3 bench.py --steps 50 --blocks 3 --no-extra --no-cpu-baseline 2>/dev/null > $O/no_collectives.json
python3 -c "$show" $O/no_collectives.json "no-collectives"
