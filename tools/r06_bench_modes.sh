set -o pipefail
O=gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
tools/profile_run.sh gpurun_out/prof_r06_c128 > $O/prof_c128.log 2>&1 && echo prof_c128_done && \
tools/profile_run.sh gpurun_out/prof_r06_c3 --n 32768 --nfreq 4096 --dtype c64 --batch 1 > $O/prof_c3.log 2>&1 && echo prof_c3_done && \
python3 bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.stderr && cp bench_detail.json $O/bench_n1_default_detail.json && \
python3 bench.py --sweeps > $O/bench_n1_sweeps.json 2> $O/bench_n1_sweeps.stderr && cp bench_detail.json $O/bench_n1_sweeps_detail.json && \
python3 bench.py --gpus 1 --in-process > $O/bench_inproc_n1.json 2> $O/bench_inproc_n1.stderr && cp bench_detail.json $O/bench_inproc_n1_detail.json && \
python3 bench.py --gpus 2 --in-process --in-process-devices 0,0 --steps 20 > $O/bench_inproc_two_workers_one_gpu.json 2> $O/bench_inproc_two.stderr && cp bench_detail.json $O/bench_inproc_two_workers_one_gpu_detail.json && \
CAF_BENCH_REHEARSE_ON_ONE_GPU=1 python3 bench.py --gpus 2 --steps 20 --blocks 2 > $O/bench_n2_rehearsal.json 2> $O/bench_n2_rehearsal.stderr && cp bench_detail.json $O/bench_n2_rehearsal_detail.json && \
CAF_BENCH_FORCE_COLLECTIVES=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --steps 20 --no-extra > $O/bench_1rank_under_rccl.json 2> $O/bench_1rank_under_rccl.stderr && \
(CAF_BENCH_REHEARSE_ON_ONE_GPU=1 CAF_BENCH_TEST_STALL=rank=1,phase=timed,seconds=200 CAF_BENCH_PHASE_LIMITS=timed=12 python3 bench.py --gpus 2 --in-process-devices 0,0 --steps 20 --blocks 2 --no-extra > $O/bench_fallback_selflaunch.json 2> $O/bench_fallback_selflaunch.stderr; echo "rc=$?" >> $O/bench_fallback_selflaunch.stderr) && \
(CAF_BENCH_REHEARSE_ON_ONE_GPU=1 CAF_BENCH_TEST_STALL=rank=1,phase=timed,seconds=200 CAF_BENCH_PHASE_LIMITS=timed=12 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 2 --in-process-devices 0,0 --steps 20 --blocks 2 --no-extra > $O/bench_fallback_external_torchrun.json 2> $O/bench_fallback_external_torchrun.stderr; echo "rc=$?" >> $O/bench_fallback_external_torchrun.stderr) && \
wc -c $O/*.json
