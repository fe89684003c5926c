"""bench.py's own control paths ON the GPU box (the CPU suite covers them with fabricated measurements): the N > 1 code path
rehearsed with two ranks sharing this one GPU over gloo, the same with one rank asleep inside the timed loop (the watchdog
must end every rank, non-zero, without a line), and the one-process path (`--in-process`) with two workers on the one GPU.
Child processes only: this process never touches the GPU for them."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra, timeout=400):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "CAF_HIP_LIB")}
    env.update(env_extra)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def _lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_two_rank_rehearsal_on_one_gpu():
    """launcher -> torchrun -> two ranks on cuda:0 -> row shards [0,200) / [200,400) of 2 x 32 surfaces per step -> peak reduction per
    step (gloo) -> ONE line from rank 0 with the contract's keys, both ranks' devices and kernel times."""
    r = _run(["--gpus", "2", "--batch", "32", "--steps", "10", "--blocks", "1", "--no-extra", "--cpu-seconds", "0.5"],
             {"CAF_BENCH_REHEARSE_ON_ONE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    (line,) = _lines(r.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak" and line["config"]["surfaces_per_step"] == 64
    assert line["config"]["rows_per_gpu"] == 200 and line["config"]["parallelism"] == "doppler-row-shard x2"
    assert [d["rank"] for d in line["config"]["rank_devices"]] == [0, 1] and len(line["extra"]["rank_kernel_ms"]) == 2
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and "phase_seconds" in line["extra"]


def test_a_stalled_rank_on_the_gpu_path_ends_the_run():
    """the same run with rank 1 asleep inside the timed loop and the phase limit at 12 s: both ranks leave with status 3 after
    naming rank, device and phase; no JSON line; well inside the bound."""
    t0 = time.time()
    r = _run(["--gpus", "2", "--batch", "32", "--steps", "10", "--blocks", "0", "--no-extra", "--no-cpu-baseline"],
             {"CAF_BENCH_REHEARSE_ON_ONE_GPU": "1", "CAF_BENCH_TEST_STALL": "rank=1,phase=timed,seconds=200",
              "CAF_BENCH_PHASE_LIMITS": "timed=12"}, timeout=300)
    assert r.returncode != 0 and not _lines(r.stdout), r.stdout
    assert "rank 0 (device cuda:0) did not finish phase 'timed' within 12 s" in r.stderr, r.stderr[-1500:]
    assert "rank 1 (device cuda:0) did not finish phase 'timed' within 12 s" in r.stderr
    assert time.time() - t0 < 200


def test_in_process_two_workers_one_gpu():
    """`--in-process` with device ids 0,0: caf_multi_surface_run_batch over two workers (host join: RCCL needs one rank per GPU),
    the contract line for BASELINE configs[1]."""
    r = _run(["--gpus", "2", "--in-process", "--in-process-devices", "0,0", "--batch", "16", "--steps", "5", "--blocks", "1", "--no-extra",
              "--cpu-seconds", "0.5"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    (line,) = _lines(r.stdout)
    assert line["metric"].startswith("CAF surfaces/sec (400 freqs x 8192 samp, c128)") and line["n_gpus"] == 1
    assert line["config"]["surfaces_per_step"] == 32 and line["config"]["rows_per_gpu"] == 200 and "host join" in line["config"]["peak_exchange"]
    assert line["extra"]["forms"]["host_join"]["planted_peaks_found"] and len(line["extra"]["rank_kernel_ms"]) == 2
