"""bench.py's control path on CPU (no GPU): `--gpus N` outside torchrun must start the N ranks as a
CHILD process tree, rendezvous, reduce shard peaks with the product's reduce_global_peak and relay
rank 0's single JSON line; a --gpus / WORLD_SIZE mismatch must exit non-zero."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_self_launch_two_ranks_plumbing_only():
    for method in ("allreduce", "allgather"):
        r = _run(["--gpus", "2", "--steps", "3", "--plumbing-only", "--peak-reduce", method])
        assert r.returncode == 0, r.stderr[-2000:]
        lines = _json_lines(r.stdout)
        assert len(lines) == 1, r.stdout                     # exactly ONE line, from rank 0
        assert lines[0]["n_gpus"] == 2 and lines[0]["plumbing_only"] and lines[0]["config"]["peak_exchange"].startswith(method)
        # the N > 1 line carries BOTH multi-GPU decompositions of the other configs: configs[3] as Doppler-row
        # shards of one surface + peak reduction, configs[4] as whole surfaces round-robin over the ranks
        ex = lines[0]["extra"]
        assert ex["configs3_c64_sharded"]["global_peak_correct"] and ex["configs3_c64_sharded"]["rows_rank0"] == 2048
        sp = ex["configs4_stream_surface_parallel"]
        assert sp["pairs_total"] == 1000 and sp["pairs_rank0"] == 500 and abs(sp["elapsed_ms_max_over_ranks"] - 2.0) < 1e-9


def test_single_rank_plumbing_needs_no_launcher():
    r = _run(["--plumbing-only", "--steps", "2", "--no-cpu-baseline"])
    assert r.returncode == 0 and _json_lines(r.stdout)[0]["n_gpus"] == 1


def test_n2_line_has_the_keys_of_the_n1_line():
    """Every bench line is put together by bench.assemble_line -- the measured run at any N and this rehearsal with
    fabricated measurements -- so the N = 2 line must carry exactly the N = 1 line's keys (the contract's keys + `roofline`
    + `cpu_baseline` + `extra`), a `cpu_baseline` object measured on rank 0 at N = 2 too, a `traffic` field that is null only
    with a stated reason, and the per-rank evidence that every rank worked."""
    sys.path.insert(0, str(ROOT))
    import bench
    r1 = _run(["--plumbing-only", "--steps", "2", "--cpu-seconds", "0.5"])
    r2 = _run(["--gpus", "2", "--plumbing-only", "--steps", "2", "--cpu-seconds", "0.5"])
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr[-1000:] + r2.stderr[-1000:]
    l1, l2 = _json_lines(r1.stdout)[0], _json_lines(r2.stdout)[0]
    assert set(l1) == set(l2) == set(bench.LINE_KEYS) | {"plumbing_only"}
    assert set(l1["roofline"]) == set(l2["roofline"]) == set(bench.ROOFLINE_KEYS)
    assert set(l1["config"]) == set(l2["config"])
    for line in (l1, l2):
        cb = line["cpu_baseline"]
        assert cb and cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb and cb["unit"] == "surfaces/s"
        roof = line["roofline"]
        assert roof["traffic"] is not None or (roof["traffic_source"] or "").startswith("none:")
        assert line["extra"]["headline_blocks"]["blocks"] >= 1
    assert set(bench.MULTI_EXTRA_KEYS) <= set(l2["extra"])
    assert len(l2["extra"]["rank_kernel_ms"]) == 2 and l2["extra"]["rccl_world"]["world_size"] == 2
    assert l2["config"]["parallelism"] == "doppler-row-shard x2" and l1["config"]["parallelism"] == "single"


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--plumbing-only"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr
    r = _run(["--gpus", "1", "--plumbing-only"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode == 2


def test_algorithmic_bytes_match_survey_8d():
    sys.path.insert(0, str(ROOT))
    import bench
    # one surface per launch, freq list included
    assert bench.algorithmic_bytes(1, 400, 4096, "c128") == 26_355_072
    assert bench.algorithmic_bytes(1, 400, 4096, "c64") == 13_180_736
    assert bench.algorithmic_bytes(1, 4096, 32768, "c64") == 1_074_348_032
    # per additional surface of a batched launch (the freq list is read once per launch)
    per128 = bench.algorithmic_bytes(2, 400, 4096, "c128") - bench.algorithmic_bytes(1, 400, 4096, "c128")
    per64 = bench.algorithmic_bytes(2, 400, 4096, "c64") - bench.algorithmic_bytes(1, 400, 4096, "c64")
    assert per128 == 65_536 + 65_536 + 26_214_400 + 6_400
    assert per64 == 2 * 32_768 + 13_107_200 + 400 * 12
    big = bench.algorithmic_bytes(2, 4096, 32768, "c64") - bench.algorithmic_bytes(1, 4096, 32768, "c64")
    assert big == 2 * 262_144 + 4096 * 65536 * 4 + 4096 * 12
    assert len(bench.kernel_source_hash()) == 16
    # per-kernel: the streaming header is not part of the batched row kernels' sources
    h = {k: bench.kernel_source_hash(k) for k in ("caf::k_seq_rows<double, 15, caf::SeqIo<double> >", "caf::k_duo_rows<float, caf::DuoIo<float> >",
                                                   "caf::k_chain_rows<float, 14, 4, 1, 0>", "")}
    assert len(set(h.values())) == 4


def test_a_stalled_rank_ends_the_run_with_a_status_and_without_a_line():
    """The join that cannot be skipped (mod.rs:452-457) must not be able to hang the bench: rank 1 sleeps inside the timed
    loop, rank 0 sits in the collective waiting for it; both watchdogs fire at the phase limit, every rank writes one stderr
    line naming its rank, device and phase, the run ends non-zero well inside the bound and NO JSON line is printed (the
    headline was never measured)."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "4", "--plumbing-only", "--no-cpu-baseline"],
             {"CAF_BENCH_TEST_STALL": "rank=1,phase=timed,seconds=150", "CAF_BENCH_PHASE_LIMITS": "timed=6"}, timeout=200)
    took = time.time() - t0
    assert r.returncode != 0, r.stdout
    assert not _json_lines(r.stdout), r.stdout
    assert "did not finish phase 'timed' within 6 s" in r.stderr and "without a result line" in r.stderr, r.stderr[-1500:]
    assert "rank 1 (device cpu)" in r.stderr or "rank 0 (device cpu)" in r.stderr
    assert took < 100, took
    # the same run without the stall passes and carries who-sits-where and the phase clock
    r = _run(["--gpus", "2", "--steps", "4", "--plumbing-only", "--no-cpu-baseline"], {"CAF_BENCH_PHASE_LIMITS": "timed=60"})
    assert r.returncode == 0, r.stderr[-1500:]
    line = _json_lines(r.stdout)[0]
    assert [d["rank"] for d in line["config"]["rank_devices"]] == [0, 1]
    assert "timed" in line["extra"]["phase_seconds"] and "rank_kernel_ms_spread" in line["extra"]
    assert "rank_kernel_ms_flag" in line["extra"]          # the fabricated kernel times (0.5 / 1.0 ms) differ by 100 %


def test_self_launch_is_refused_under_a_profiler_preload():
    """rocprofv3's preloaded tool library initialises the GPU before bench.py's first line: starting torchrun from such a
    process is the exec-after-GPU-init this pool forbids.  --in-process / --emulate-rank-of are the ways to profile."""
    r = _run(["--gpus", "2", "--plumbing-only"], {"ROCPROFILER_TOOL_TEST_MARKER": "1"})
    assert r.returncode == 2 and "--emulate-rank-of" in r.stderr and not _json_lines(r.stdout)


def test_watchdog_prints_the_measured_headline_before_leaving():
    """A hang AFTER the timed region costs the run its status but not the measurement: the phase's on_expiry callback runs under
    the line lock (it prints the headline snapshot once), then the process writes its stderr line and leaves with status 3; a
    phase without a callback leaves without any line."""
    code = r'''
import sys, time
sys.path.insert(0, %r)
import bench_common as bc
bc.guard_stdout()
wd = bc.PhaseWatchdog(3, "cuda:3")
wd.limits["extras"] = 1.0
def cb():
    if not wd.printed:
        wd.printed = True
        bc.emit_line({"value": 1.0, "extra": {"error": "extras hung"}})
wd.enter("setup"); wd.enter("timed"); wd.enter("extras", on_expiry=cb if sys.argv[1] == "cb" else None)
time.sleep(30)
''' % str(ROOT)
    for mode, want_line in (("cb", True), ("none", False)):
        r = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=60)
        assert r.returncode == 3, (r.returncode, r.stderr)
        assert "rank 3 (device cuda:3) did not finish phase 'extras' within 1 s" in r.stderr
        lines = _json_lines(r.stdout)
        assert (len(lines) == 1 and lines[0]["extra"]["error"] == "extras hung") if want_line else not lines
        assert ("without a result line" in r.stderr) == (not want_line)


def test_four_ranks_plumbing_only():
    """world size 4 (400 rows -> 100 per rank; 4096 rows -> 1024; 1000 pairs -> 250): the launcher, the rendezvous, both peak
    reductions and the per-rank bookkeeping of the line at a world size that is neither 1 nor 2."""
    r = _run(["--gpus", "4", "--steps", "3", "--plumbing-only", "--no-cpu-baseline"], timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    (line,) = _json_lines(r.stdout)
    assert line["n_gpus"] == 4 and line["config"]["rows_per_gpu"] == 100 and line["config"]["surfaces_per_step"] == 4 * 256
    assert [d["rank"] for d in line["config"]["rank_devices"]] == [0, 1, 2, 3] and len(line["extra"]["rank_kernel_ms"]) == 4
    assert line["extra"]["configs3_c64_sharded"]["rows_rank0"] == 1024 and line["extra"]["configs4_stream_surface_parallel"]["pairs_rank0"] == 250
