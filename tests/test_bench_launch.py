"""bench.py's control path on CPU (no GPU): `--gpus N` outside torchrun must start the N ranks as a
CHILD process tree, rendezvous, reduce shard peaks with the product's reduce_global_peak and relay
rank 0's single JSON line; the line must stay small and strictly parseable (round 5's grew to 21 KB and
the driver could not recover it); a --gpus / WORLD_SIZE mismatch must exit non-zero; a run that fails
before the headline must fall back to the one-process path as a fresh child."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
_RANK_VARS = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "CAF_BENCH_UNDER_LAUNCHER")


def _env(env_extra, detail):
    env = {k: v for k, v in os.environ.items() if k not in _RANK_VARS}
    env["CAF_BENCH_DETAIL"] = str(detail)
    env.update(env_extra or {})
    return env


def _run(args, env_extra=None, timeout=240, tmp=None):
    detail = Path(tmp or "/tmp") / f"bench_detail_test_{os.getpid()}.json"
    if detail.exists():
        detail.unlink()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=_env(env_extra, detail),
                       timeout=timeout, cwd=ROOT)
    r.detail = json.loads(detail.read_text()) if detail.exists() else None
    return r


def _strict(text):
    """json.loads that refuses what is not JSON (NaN, Infinity): what a parser other than Python's would refuse"""
    def bad(name):
        raise ValueError(f"non-JSON constant {name}")
    return json.loads(text, parse_constant=bad)


def _json_lines(out):
    return [_strict(l) for l in out.splitlines() if l.startswith("{")]


def _the_line(r):
    """the contract: stdout is EXACTLY one line, that line is strict JSON and at most LINE_LIMIT bytes"""
    sys.path.insert(0, str(ROOT))
    import bench_common
    assert r.stdout.endswith("\n") and r.stdout.count("\n") == 1, r.stdout[-2000:]
    assert len(r.stdout.encode()) <= bench_common.LINE_LIMIT, len(r.stdout.encode())
    return _strict(r.stdout)


def test_self_launch_two_ranks_plumbing_only(tmp_path):
    for method in ("allreduce", "allgather"):
        r = _run(["--gpus", "2", "--steps", "3", "--plumbing-only", "--peak-reduce", method], tmp=tmp_path)
        assert r.returncode == 0, r.stderr[-2000:]
        line = _the_line(r)                                   # exactly ONE line, from rank 0, relayed by the launcher
        assert line["n_gpus"] == 2 and line["plumbing_only"] and line["config"]["peak_exchange"].startswith(method)
        # the N > 1 record carries BOTH multi-GPU decompositions of the other configs: configs[3] as Doppler-row
        # shards of one surface + peak reduction, configs[4] as whole surfaces round-robin over the ranks
        assert line["extra"]["configs3_c64_sharded"]["global_peak_correct"] is True
        ex = r.detail["extra"]
        assert ex["configs3_c64_sharded"]["global_peak_correct"] and ex["configs3_c64_sharded"]["rows_rank0"] == 2048
        sp = ex["configs4_stream_surface_parallel"]
        assert sp["pairs_total"] == 1000 and sp["pairs_rank0"] == 500 and abs(sp["elapsed_ms_max_over_ranks"] - 2.0) < 1e-9
        assert line["extra"]["detail_file"] and Path(line["extra"]["detail_file"]).exists()


def test_single_rank_plumbing_needs_no_launcher(tmp_path):
    r = _run(["--plumbing-only", "--steps", "2", "--no-cpu-baseline"], tmp=tmp_path)
    assert r.returncode == 0 and _the_line(r)["n_gpus"] == 1


@pytest.mark.parametrize("mode", ["world1", "world2", "in_process2", "in_process8"])
def test_the_line_is_one_small_strict_json_line(mode, tmp_path):
    """the LAST stdout line is the contract line and nothing else -- one line, strict JSON, <= 4 096 bytes, at
    world 1, world 2 and on the one-process path (8 workers: the longest rank lists the line will ever carry); the full
    record is in the detail file, the same record is on stderr, and the line's figures are the record's."""
    args = {"world1": [], "world2": ["--gpus", "2"], "in_process2": ["--gpus", "2", "--in-process"],
            "in_process8": ["--gpus", "8", "--in-process"]}[mode]
    r = _run([*args, "--plumbing-only", "--steps", "2", "--cpu-seconds", "0.5"], tmp=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _the_line(r)
    sys.path.insert(0, str(ROOT))
    import bench
    assert set(line) == set(bench.LINE_KEYS) | {"plumbing_only"}
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"]
    cb = line["cpu_baseline"]
    assert set(cb) == {"value", "unit", "cores", "kind", "ms_per_surface", "single_thread_ms_per_surface", "host_cpu", "sample"}
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    # no prose in the line; the prose and the tables are in the record
    assert "traffic_is" not in line["roofline"] and "flavours" not in cb and "corresponds_to" not in cb
    assert "flavours" in r.detail["cpu_baseline"] and "traffic_is" in r.detail["roofline"]
    for k in bench.RANK_EVIDENCE_KEYS:
        assert k in line["config"], k
    n = {"world1": 1, "world2": 2, "in_process2": 2, "in_process8": 8}[mode]
    assert len(line["config"]["rank_devices"]) == n and len(line["config"]["rank_kernel_ms"]) == n
    # the stderr copy of the full record parses too and equals the file
    det = [l for l in r.stderr.splitlines() if l.startswith("bench.py detail: ")]
    assert len(det) == 1 and _strict(det[0][len("bench.py detail: "):]) == r.detail
    assert r.detail["cpu_baseline"]["value"] == pytest.approx(cb["value"], rel=1e-5)
    # ... and the LAST stderr line is a copy of the stdout line: a capture that appends stderr to stdout and keeps 8 KB of the
    # end (the driver's record does) still shows the whole line, not only the end of the long detail record
    merged_tail = (r.stdout + r.stderr).encode()[-8192:].decode(errors="replace")
    assert "bench.py line: " + r.stdout.strip() in merged_tail and r.stderr.rstrip().splitlines()[-1] == "bench.py line: " + r.stdout.strip()


def test_a_line_that_would_be_too_long_loses_extras_not_contract_keys():
    sys.path.insert(0, str(ROOT))
    import bench_common as bc
    res = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "w" * 5000},
           "roofline": {"bound": "hbm", "achieved": float("nan"), "frac": 0.5}, "cpu_baseline": {"value": 2.0, "sample": "s" * 5000},
           "extra": {"configs2_c64": {"value": 1.0, "frac": 0.1}, "error": "e" * 9000, "phase_seconds": {f"p{i}": 1.0 for i in range(400)}}}
    line = bc.shrink_to_limit(bc.compact_line(bc.sanitize(res)))
    text = json.dumps(line, allow_nan=False)
    assert len(text) + 1 <= bc.LINE_LIMIT
    assert line["value"] == 1.0 and line["roofline"]["frac"] == 0.5 and line["roofline"]["achieved"] is None and line["cpu_baseline"]["value"] == 2.0
    assert "phase_seconds" in line["extra"]["dropped"]


def test_n2_line_has_the_keys_of_the_n1_line(tmp_path):
    """Every bench record is put together by bench.assemble_line -- the measured run at any N and this rehearsal with
    fabricated measurements -- so the N = 2 line must carry exactly the N = 1 line's keys (the contract's keys + `roofline`
    + `cpu_baseline` + `extra`), a `cpu_baseline` object measured on rank 0 at N = 2 too, a `traffic` field that is null only
    with a stated reason, and -- in `config`, which the driver's record keeps whole -- the per-rank evidence that every rank worked."""
    sys.path.insert(0, str(ROOT))
    import bench
    r1 = _run(["--plumbing-only", "--steps", "2", "--cpu-seconds", "0.5"], tmp=tmp_path)
    d1 = r1.detail
    r2 = _run(["--gpus", "2", "--plumbing-only", "--steps", "2", "--cpu-seconds", "0.5"], tmp=tmp_path)
    d2 = r2.detail
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr[-1000:] + r2.stderr[-1000:]
    l1, l2 = _the_line(r1), _the_line(r2)
    assert set(l1) == set(l2) == set(bench.LINE_KEYS) | {"plumbing_only"}
    assert set(l1["roofline"]) == set(l2["roofline"])
    assert set(d1["roofline"]) == set(d2["roofline"]) == set(bench.ROOFLINE_KEYS)
    assert set(l1["config"]) == set(l2["config"]) and set(d1["config"]) == set(d2["config"])
    for line in (l1, l2):
        cb = line["cpu_baseline"]
        assert cb and cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb and cb["unit"] == "surfaces/s"
        roof = line["roofline"]
        assert roof["traffic"] is not None or (roof["traffic_source"] or "").startswith("none:")
        assert line["extra"]["headline_blocks"]["blocks"] >= 1
    assert set(bench.MULTI_EXTRA_KEYS) <= set(l2["extra"])
    assert len(l2["config"]["rank_kernel_ms"]) == 2 and l2["config"]["rccl_world"] == {"world_size": 2, "backend": "gloo"}
    assert l1["config"]["rccl_world"] is None and len(l1["config"]["rank_kernel_ms"]) == 1
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


def test_a_stalled_rank_ends_the_run_with_a_status_and_without_a_line(tmp_path):
    """The join that cannot be skipped (mod.rs:452-457) must not be able to hang the bench: rank 1 sleeps inside the timed
    loop, rank 0 sits in the collective waiting for it; both watchdogs fire at the phase limit, every rank writes one stderr
    line naming its rank, device and phase, the run ends non-zero well inside the bound and -- with --no-fallback -- NO JSON
    line is printed (the headline was never measured)."""
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "4", "--plumbing-only", "--no-cpu-baseline", "--no-fallback"],
             {"CAF_BENCH_TEST_STALL": "rank=1,phase=timed,seconds=150", "CAF_BENCH_PHASE_LIMITS": "timed=6"}, timeout=200, tmp=tmp_path)
    took = time.time() - t0
    assert r.returncode != 0, r.stdout
    assert not _json_lines(r.stdout), r.stdout
    assert "did not finish phase 'timed' within 6 s" in r.stderr and "without a result line" in r.stderr, r.stderr[-1500:]
    assert "rank 1 (device cpu)" in r.stderr or "rank 0 (device cpu)" in r.stderr
    assert took < 100, took
    # the same run without the stall passes and carries who-sits-where and the phase clock
    r = _run(["--gpus", "2", "--steps", "4", "--plumbing-only", "--no-cpu-baseline"], {"CAF_BENCH_PHASE_LIMITS": "timed=60"}, tmp=tmp_path)
    assert r.returncode == 0, r.stderr[-1500:]
    line = _the_line(r)
    assert [d["rank"] for d in line["config"]["rank_devices"]] == [0, 1]
    assert "timed" in line["extra"]["phase_seconds"] and line["config"]["rank_kernel_ms_spread"] == pytest.approx(1.0)
    assert line["config"]["rank_kernel_ms_flag"]           # the fabricated kernel times (0.5 / 1.0 ms) differ by 100 %
    assert "fallback_from" not in line["config"]


def test_a_failed_torchrun_tree_falls_back_to_the_one_process_path(tmp_path):
    """first multi-GPU contact must yield a number even if one path fails.  The launcher (which has touched
    neither torch nor HIP) sees the torchrun tree end non-zero without a line -- rank 1 asleep in the timed loop, both
    watchdogs fire -- and starts `bench.py --gpus 2 --in-process` as a SECOND fresh child; that child's line is relayed with
    config.fallback_from naming the path that failed, its status and the tail of its stderr; exit status 0."""
    r = _run(["--gpus", "2", "--steps", "4", "--plumbing-only", "--no-cpu-baseline"],
             {"CAF_BENCH_TEST_STALL": "rank=1,phase=timed,seconds=150", "CAF_BENCH_PHASE_LIMITS": "timed=5"}, timeout=240, tmp=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _the_line(r)
    fb = line["config"]["fallback_from"]
    assert fb["path"] == "torchrun" and fb["rc"] not in (0, None) and "did not finish phase 'timed'" in fb["stderr_tail"]
    assert line["n_gpus"] == 2 and "in-process" in line["config"]["parallelism"] and line["plumbing_only"]
    assert [d["worker"] for d in line["config"]["rank_devices"]] == [0, 1]
    assert "running the same headline through the one-process path" in r.stderr


def test_a_failed_correctness_gate_is_never_answered_by_the_fallback(tmp_path):
    """A WRONG ANSWER on the torchrun path must fail the run: neither layer starts the one-process path to print another path's
    number over it.  (The gate is forced to fail on every rank; the ranks print a marker the launcher reads.)"""
    r = _run(["--gpus", "2", "--steps", "2", "--plumbing-only", "--no-cpu-baseline"], {"CAF_BENCH_TEST_FAIL_GATE": "1"}, timeout=240, tmp=tmp_path)
    assert r.returncode != 0 and not r.stdout, (r.returncode, r.stdout)
    assert "CORRECTNESS GATE FAILED on rank" in r.stderr and "running the same headline through the one-process path" not in r.stderr


def test_under_an_external_torchrun_rank0_falls_back_and_every_rank_leaves_with_its_status(tmp_path):
    """The driver starts torchrun itself (RANK / WORLD_SIZE set: bench.py's own launcher is not involved).  There a failure
    before the headline is answered by rank 0: its watchdog starts the one-process path as a fresh child and relays the
    line; rank 1 waits for rank 0's verdict instead of leaving at once (the launcher would end rank 0 with it).  torchrun
    exits 0 and its stdout holds the one line."""
    detail = tmp_path / "detail.json"
    port = subprocess.run([sys.executable, "-c", "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])"],
                          capture_output=True, text=True).stdout.strip()
    env = _env({"CAF_BENCH_TEST_STALL": "rank=1,phase=timed,seconds=150", "CAF_BENCH_PHASE_LIMITS": "timed=5", "CAF_BENCH_FALLBACK_SETTLE_S": "1",
                "OMP_NUM_THREADS": "1"}, detail)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--plumbing-only", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    (line,) = _json_lines(r.stdout)
    fb = line["config"]["fallback_from"]
    assert fb["path"].startswith("torchrun (external") and "did not finish phase 'timed'" in fb["stderr_tail"]
    assert "rank 1" in r.stderr and "waiting for rank 0's one-process fallback" in r.stderr
    # and with --no-fallback the same run ends non-zero without a line
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--plumbing-only", "--no-cpu-baseline",
                        "--no-fallback"], capture_output=True, text=True, env=env, timeout=240, cwd=ROOT)
    assert r.returncode != 0 and not _json_lines(r.stdout)


def test_a_rank_that_raises_alone_waits_for_rank0s_fallback_under_an_external_torchrun(tmp_path):
    """The other way a run fails before the headline: rank 1 RAISES inside the timed loop (a device fault, an allocation) while
    rank 0 sits in the collective.  Rank 1's handler does not leave -- it waits for rank 0's verdict; rank 0's watchdog runs into
    the phase limit, starts the one-process path, relays the line; both ranks leave with status 0."""
    detail = tmp_path / "detail.json"
    port = subprocess.run([sys.executable, "-c", "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])"],
                          capture_output=True, text=True).stdout.strip()
    env = _env({"CAF_BENCH_TEST_STALL": "rank=1,phase=timed,raise=1", "CAF_BENCH_PHASE_LIMITS": "timed=5", "CAF_BENCH_FALLBACK_SETTLE_S": "1",
                "OMP_NUM_THREADS": "1"}, detail)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--plumbing-only", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    (line,) = _json_lines(r.stdout)
    assert line["config"]["fallback_from"]["path"].startswith("torchrun (external") and line["plumbing_only"]
    assert "rank 1 failed before the headline: RuntimeError" in r.stderr and "waiting for rank 0's one-process fallback" in r.stderr


def test_self_launch_is_refused_under_a_profiler_preload():
    """rocprofv3's preloaded tool library initialises the GPU before bench.py's first line: starting torchrun from such a
    process is the exec-after-GPU-init this pool forbids.  --in-process / --emulate-rank-of are the ways to profile.
    Detected by the preload itself (what rocprofv3 sets: ROCP_TOOL_LIBRARIES, LD_PRELOAD of its tool library), NOT by any
    ROCPROF* variable a shell may carry: ROCPROFILER_LOG_LEVEL alone must not stop a run."""
    r = _run(["--gpus", "2", "--plumbing-only"], {"ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"})
    assert r.returncode == 2 and "--emulate-rank-of" in r.stderr and not _json_lines(r.stdout)
    r = _run(["--gpus", "2", "--plumbing-only", "--steps", "2", "--no-cpu-baseline"], {"ROCPROFILER_LOG_LEVEL": "warning", "ROCPROF_OUTPUT_PATH": "/tmp/x"})
    assert r.returncode == 0 and _the_line(r)["n_gpus"] == 2, r.stderr[-1500:]


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


def test_four_ranks_plumbing_only(tmp_path):
    """world size 4 (400 rows -> 100 per rank; 4096 rows -> 1024; 1000 pairs -> 250): the launcher, the rendezvous, both peak
    reductions and the per-rank bookkeeping of the line at a world size that is neither 1 nor 2."""
    r = _run(["--gpus", "4", "--steps", "3", "--plumbing-only", "--no-cpu-baseline"], timeout=300, tmp=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _the_line(r)
    assert line["n_gpus"] == 4 and line["config"]["rows_per_gpu"] == 100 and line["config"]["surfaces_per_step"] == 4 * 256
    assert [d["rank"] for d in line["config"]["rank_devices"]] == [0, 1, 2, 3] and len(line["config"]["rank_kernel_ms"]) == 4
    ex = r.detail["extra"]
    assert ex["configs3_c64_sharded"]["rows_rank0"] == 1024 and ex["configs4_stream_surface_parallel"]["pairs_rank0"] == 250


def test_the_committed_full_records_compact_to_small_strict_lines():
    """The full records of real runs (profiles/r06_misc/*_detail.json: default, --sweeps, --in-process, the N = 2 rehearsal) through
    the same compaction as a live run: each line <= 4 096 bytes, strict JSON, with value / roofline.frac / cpu_baseline.value and
    one scalar pair per other config -- a regression guard that needs no GPU (the --sweeps record is 22 KB; its line 2.6 KB)."""
    sys.path.insert(0, str(ROOT))
    import bench_common as bc
    files = sorted((ROOT / "profiles" / "r06_misc").glob("*_detail.json"))
    assert len(files) >= 4
    for f in files:
        res = json.loads(f.read_text())
        line = bc.shrink_to_limit(bc.compact_line(bc.sanitize(res)))
        text = json.dumps(line, allow_nan=False)
        assert len(text) + 1 <= bc.LINE_LIMIT, (f.name, len(text))
        back = _strict(text)
        assert back["value"] == res["value"] and back["roofline"]["frac"] == pytest.approx(res["roofline"]["frac"], rel=1e-5)
        assert back["cpu_baseline"]["value"] == pytest.approx(res["cpu_baseline"]["value"], rel=1e-5) and "dropped" not in back["extra"]
        for k, v in back["extra"].items():
            if isinstance(v, dict):
                assert all(not isinstance(x, (dict, list)) for x in v.values()), (f.name, k)   # scalars only: nothing nested in the line
    sweeps = json.loads((ROOT / "profiles" / "r06_misc" / "bench_n1_sweeps_detail.json").read_text())
    assert len(sweeps["extra"]["configs4_stream"]["forms"]) == 13 and len(json.dumps(sweeps)) > 20000


def test_stream_sweep_tool_parses_and_analyses_a_trace(tmp_path):
    """tools/stream_sweep.py (fourteen round-2..5 scripts in one): its CSV analyses run without a GPU; the GPU commands at least
    parse their arguments (presets, forms)."""
    tool = str(ROOT / "tools" / "stream_sweep.py")
    csvf = tmp_path / "kernel_trace.csv"
    rows = ["Kernel_Name,Start_Timestamp,End_Timestamp,Queue_Id"]
    t = 1000
    for i in range(40):
        rows.append(f"caf::k_seq_prepare<double>(args),{t},{t + 16000},1")
        rows.append(f"caf::k_seq_rows<double>(args),{t + 17000},{t + 17000 + 290000},{1 + i % 2}")
        t += 300000
    csvf.write_text("\n".join(rows) + "\n")
    r = subprocess.run([sys.executable, tool, "busy", str(csvf)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "row kernels over" in r.stdout and "at least one row kernel running" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, tool, "trace", str(csvf)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "k_seq_rows" in r.stdout and "median" in r.stdout
    r = subprocess.run([sys.executable, tool, "rates", "--help"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "--preset" in r.stdout
    sys.path.insert(0, str(ROOT / "tools"))
    import stream_sweep
    assert stream_sweep.parse_form("20:2:m") == dict(batch=20, nslots=2, split=False, one_kernel=False, two_kernels=False, three_kernels=False, memcpy_nodes=True)
    assert stream_sweep.parse_form("4:2:s2")["split"] and stream_sweep.parse_form("4:2:s2")["two_kernels"]
    assert set(stream_sweep.PRESETS) == {"1000", "stability", "native", "batch", "slots", "fixed", "probe", "reserve"}
