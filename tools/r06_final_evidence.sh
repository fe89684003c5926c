#!/bin/bash
# Round-6 evidence pass on the round's final sources (GPU box, repo root): one full GPU suite, the driver's bench command,
# the kernel stats of the one-process path, the bit-exactness soaks.  Steps are joined with &&: nothing starts after a failure.
set -o pipefail
O=gpurun_out/r06_final; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 && tail -1 $O/gpu_tests.log && \
(time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.stderr) 2> $O/bench_driver_command_time.txt && cp bench_detail.json $O/bench_driver_command_detail.json && wc -c $O/bench_driver_command.json && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/inproc_stats -- python3 bench.py --gpus 1 --in-process --steps 20 --warmup 3 --blocks 0 --no-cpu-baseline --no-extra > $O/inproc_bench_under_rocprof.json 2> /dev/null && echo inproc_profile_done && \
timeout -k 10 130 python tools/soak.py 100 > $O/soak_batched_vs_single.txt 2>&1 && tail -2 $O/soak_batched_vs_single.txt && \
timeout -k 10 200 python tools/stream_sweep.py soak 4096 12 > $O/soak_stream.txt 2>&1 && tail -1 $O/soak_stream.txt && \
timeout -k 10 130 python tools/host_api_soak.py 100000 > $O/soak_host_api.txt 2>&1 && tail -1 $O/soak_host_api.txt && \
timeout -k 10 150 python tools/multi_batch_soak.py 100 > $O/soak_multi_batch.txt 2>&1 && tail -1 $O/soak_multi_batch.txt
