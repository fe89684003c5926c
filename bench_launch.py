"""bench_launch.py -- how bench.py gets its ranks started and what it does when that fails: the launcher of `--gpus N` outside
torchrun (the torchrun tree as a CHILD with its stdout captured), the one-process fallback as a second fresh child at both
launch layers (this launcher; rank 0 under a torchrun that is not ours), and the helpers they share.  Nothing here imports
torch or touches HIP: a process that has initialised the GPU is never replaced, and the launcher never initialises it."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from pathlib import Path

from bench_common import emit_line, shrink_to_limit, under_rocprofiler

BENCH = Path(__file__).resolve().parent / "bench.py"
LAUNCH_TIMEOUT_S = 480         # self_launch: the whole torchrun tree (a healthy N = 8 run: import + ~1 minute)
FALLBACK_TIMEOUT_S = 360       # the one-process fallback child


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
             "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
             "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING",
             "TORCHELASTIC_ERROR_FILE", "CAF_BENCH_UNDER_LAUNCHER")


def last_json_line(text: str):
    """the last stdout line that is a JSON object, or None"""
    import json
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def run_child(cmd, env, limit_s):
    """One child process tree in its own session: stdout captured (it carries at most the one line), stderr passed through
    and its tail kept.  Past `limit_s` the tree's process GROUP -- the one started here, nothing found by name -- is ended.
    -> (rc | None when it had to be ended, stdout text, stderr tail)."""
    import collections
    import signal
    import threading
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    tail = collections.deque(maxlen=40)
    own = collections.deque(maxlen=6)   # bench.py's own one-line diagnostics say more than a launcher's failure summary
    out = []

    def pump_err():
        for raw in iter(proc.stderr.readline, b""):
            os.write(2, raw)
            if raw.startswith(b"bench.py detail: "):
                continue
            text = raw.decode("utf-8", "replace")
            (own if "bench.py" in text and "did not finish" in text or text.startswith("bench.py:") else tail).append(text)

    def pump_out():
        out.append(proc.stdout.read())

    threads = [threading.Thread(target=pump_err, daemon=True), threading.Thread(target=pump_out, daemon=True)]
    for t in threads:
        t.start()
    rc = None
    try:
        rc = proc.wait(timeout=limit_s)
    except subprocess.TimeoutExpired:
        for sig, grace in ((signal.SIGTERM, 10), (signal.SIGKILL, 10)):
            try:
                os.killpg(proc.pid, sig)   # (start_new_session: the group id is the child's pid)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        tail.append(f"bench.py: the child did not finish within {limit_s:g} s and was ended\n")
    for t in threads:
        t.join(timeout=5)
    return rc, (out[0] if out else b"").decode("utf-8", "replace"), ("".join(own) or "".join(tail))[-1500:]


def in_process_fallback_cmd(args):
    """`bench.py --gpus N --in-process` with this run's measurement flags"""
    cmd = [sys.executable, str(BENCH), "--gpus", str(args.gpus), "--in-process", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--batch", str(args.batch), "--dtype", args.dtype, "--blocks", str(args.blocks),
           "--cpu-seconds", str(args.cpu_seconds), "--cpu-threads", str(args.cpu_threads)]
    for flag, on in (("--no-cpu-baseline", args.no_cpu_baseline), ("--no-check", args.no_check), ("--no-extra", args.no_extra),
                     ("--plumbing-only", args.plumbing_only), ("--sweeps", args.sweeps)):
        if on:
            cmd.append(flag)
    if args.in_process_devices:
        cmd += ["--in-process-devices", args.in_process_devices]
    return cmd


def run_in_process_fallback(args, failed):
    """Start the one-process path as a FRESH child (clean of every rank variable) and relay its line with
    config.fallback_from = `failed` ({path, rc, stderr_tail} of what did not produce a line).  -> True if a line was printed."""
    env = {k: v for k, v in os.environ.items() if k not in _RANK_ENV}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"bench.py: the {failed.get('path')} path gave no result (rc {failed.get('rc')}); running the same headline through the "
          "one-process path (--in-process) as a fresh child", file=sys.stderr)
    rc, out, err = run_child(in_process_fallback_cmd(args), env, FALLBACK_TIMEOUT_S)
    line = last_json_line(out)
    if line is None or (line.get("value") is None and not line.get("plumbing_only")):
        print(f"bench.py: the one-process fallback gave no result either (rc {rc})", file=sys.stderr)
        return False
    line.setdefault("config", {})["fallback_from"] = {"path": failed.get("path"), "rc": failed.get("rc"),
                                                      "stderr_tail": str(failed.get("stderr_tail", ""))[-300:]}
    if rc != 0:
        line["config"]["child_rc"] = rc
    line.setdefault("extra", {})
    emit_line(shrink_to_limit(line))
    return True


def self_launch(args) -> int:
    """--gpus N > 1 outside torchrun: run the N ranks as a child process tree.  Nothing in THIS process has imported torch or
    touched HIP (a process that initialised the GPU must never be replaced) -- unless a profiler's preloaded tool library did
    it for us: then the launch is refused.  The tree's stdout is captured; its line is relayed when it holds a measured
    headline (a later phase may have failed: the line then says so under extra.error and config.child_rc).  Otherwise the
    one-process path runs as a second fresh child (run_in_process_fallback).  Exit status 0 only if a line was printed."""
    if under_rocprofiler():
        print("bench.py: --gpus N > 1 under rocprofv3 would start torchrun from a process whose GPU the profiler's preload has "
              "already initialised; profile a rank's launch shape with --emulate-rank-of N, or the one-process path with "
              "--in-process", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(BENCH),
           *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env["CAF_BENCH_UNDER_LAUNCHER"] = "1"   # the ranks leave the fallback to this process (it holds no GPU and outlives them)
    rc, out, err = run_child(cmd, env, float(os.environ.get("CAF_BENCH_LAUNCH_TIMEOUT_S", LAUNCH_TIMEOUT_S)))
    line = last_json_line(out)
    if line is not None and (line.get("value") is not None or (line.get("plumbing_only") and rc == 0)):
        if rc != 0:
            line.setdefault("config", {})["child_rc"] = rc
        line.setdefault("extra", {})
        emit_line(shrink_to_limit(line))
        return 0
    if args.no_fallback or "CORRECTNESS GATE FAILED" in err:   # (a wrong answer is never papered over by another path's number)
        return rc if rc else 3
    return 0 if run_in_process_fallback(args, {"path": "torchrun", "rc": rc, "stderr_tail": err}) else (rc if rc else 3)


class RankFallback:
    """Launched by a torchrun that is NOT ours (RANK / WORLD_SIZE set, no CAF_BENCH_UNDER_LAUNCHER): when a phase before the
    headline fails or overruns, rank 0 starts the one-process path as a fresh child and relays its line; every other rank
    waits for rank 0's verdict (a file under /tmp keyed by the launcher's pid and port) instead of leaving at once -- the
    launcher ends the remaining ranks as soon as one exits non-zero, and rank 0 needs about a minute.  Every rank then leaves
    with rank 0's status.  Called from the watchdog thread (a main thread stuck in a collective cannot be unwound) or from
    the main thread's exception handler; whoever comes first runs it, once."""

    WAIT_S = 420.0

    def __init__(self, args, rank, world):
        import threading
        self.args, self.rank, self.world = args, rank, world
        self.enabled = (world > 1 and not args.no_fallback and os.environ.get("CAF_BENCH_UNDER_LAUNCHER") != "1"
                        and not args.in_process and not args.emulate_rank_of)
        self.flag = Path("/tmp") / f"caf_bench_fallback_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"
        self._once = threading.Lock()
        if self.enabled and rank == 0:
            try:
                self.flag.unlink()
            except OSError:
                pass

    def run(self, reason: str) -> int:
        """-> the status this rank should leave with"""
        if not self.enabled:
            return 3
        if not self._once.acquire(blocking=False):
            time.sleep(self.WAIT_S)   # the other thread of this process is running it and will end the process
            return 3
        if self.rank == 0:
            time.sleep(float(os.environ.get("CAF_BENCH_FALLBACK_SETTLE_S", "3")))   # let the other ranks reach their own limits
            ok = False
            try:
                ok = run_in_process_fallback(self.args, {"path": "torchrun (external launcher)", "rc": None, "stderr_tail": reason})
            finally:   # (whatever happened here, the other ranks must not wait out their whole patience for a verdict)
                try:
                    self.flag.write_text("ok" if ok else "fail")
                except OSError:
                    pass
            return 0 if ok else 3
        t_end = time.monotonic() + self.WAIT_S
        os.write(2, f"bench.py: rank {self.rank}: {reason}; waiting for rank 0's one-process fallback\n".encode())
        while time.monotonic() < t_end:
            try:
                return 0 if self.flag.read_text().strip() == "ok" else 3
            except OSError:
                time.sleep(0.5)
        return 3
