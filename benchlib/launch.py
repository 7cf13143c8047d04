"""`python bench.py --gpus N` without a launcher: bench.py starts its own N ranks.

The reference's distributed entry point spawns its ranks itself (tasks/node_classification_dist.py:42-61: `mp.spawn` +
`init_process_group` on a local address); the driver's documented form is `python -m torch.distributed.run ... bench.py`,
where RANK / WORLD_SIZE are already in the environment and nothing here runs.  When they are NOT (a bare
`python bench.py --gpus 8`), the parent process below

  * starts N children of the same command line, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
    MASTER_PORT=<a free port> set -- plain processes, no elastic agent, nothing imported from torch in the parent;
  * relays rank 0's stdout (the ONE JSON line) and sends every other rank's stdout to stderr;
  * ends the job when any rank fails (the others would wait in a collective for ever), prints a `value: null` line naming
    the rank, its exit code and the tail of its stderr, and exits non-zero;
  * ends a job that is still running after SGL_BENCH_LAUNCH_TIMEOUT seconds (default 5400) the same way;
  * never prints more than one line to stdout.
"""
import json
import os
import socket
import subprocess
import sys
import tempfile
import time


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _null_line(metric, n, error, **extra):
    out = {"metric": metric, "value": None, "unit": "edge·featdim/s", "n_gpus": n, "higher_is_better": True,
           "error": error}
    out.update(extra)
    return json.dumps(out)


def needs_self_launch(args, env=os.environ):
    """a bare `python bench.py --gpus N` with N > 1: no launcher has set the rank variables"""
    return args.gpus > 1 and "WORLD_SIZE" not in env and "RANK" not in env


def self_launch(args, argv, script, metric, grace_s=15.0, poll_s=0.2):
    """run `script argv` as args.gpus ranks; returns the exit code for the parent (0 iff rank 0 printed its line and every
    rank exited 0)"""
    n = args.gpus
    engine = os.environ.get("SGL_BENCH_ENGINE", "")
    if engine != "one_gpu_gloo":
        # ask the runtime for the device count in a child: the parent must not hold a HIP context of its own
        probe = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                               capture_output=True, text=True)
        have = int(probe.stdout.strip().splitlines()[-1]) if probe.returncode == 0 and probe.stdout.strip() else 0
        if have < n:
            print(_null_line(metric, n, f"--gpus {n} asked for, {have} GPU(s) visible on this node"), flush=True)
            return 2
    port = _free_port()
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SGL_BENCH_SELF_LAUNCHED="1")
    procs, errs = [], []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        err = tempfile.TemporaryFile(mode="w+")
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=env, stdout=out0 if r == 0 else err, stderr=err,
                                      start_new_session=True))
    failed = None
    limit_s = float(os.environ.get("SGL_BENCH_LAUNCH_TIMEOUT", "5400"))     # a job that hangs without any rank dying
    t_start = time.monotonic()
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad and failed is None:
                failed = bad[0]
                t_fail = time.monotonic()
            if all(c is not None for c in codes):
                break
            if failed is not None and time.monotonic() - t_fail > grace_s:
                break                                  # the survivors are waiting for the dead rank: end them
            if failed is None and time.monotonic() - t_start > limit_s:
                alive = [r for r, c in enumerate(codes) if c is None]
                failed = (alive[0], f"none: still running after {limit_s:.0f} s (SGL_BENCH_LAUNCH_TIMEOUT)")
                break
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 15)               # exactly the process groups started above
                except OSError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    pass
    out0.seek(0)
    lines = [ln for ln in out0.read().splitlines() if ln.strip().startswith("{")]
    for r, err in enumerate(errs):
        err.seek(0)
        text = err.read()
        if text:
            sys.stderr.write(f"---- rank {r} ----\n{text}\n" if n > 1 else text)
    sys.stderr.flush()
    if failed is None and lines:
        print(lines[-1], flush=True)
        return 0
    if lines:
        # rank 0 got its measured line out before something else went wrong: the line stands, the failure is recorded in it
        try:
            j = json.loads(lines[-1])
            j["launcher"] = {"failed_rank": failed[0], "exit_code": failed[1], "note": "after rank 0 had printed its line"}
            print(json.dumps(j), flush=True)
            return 0 if j.get("value") is not None else 3
        except ValueError:
            pass
    r, code = failed if failed is not None else (0, procs[0].returncode)
    errs[r].seek(0)
    tail = errs[r].read()[-600:]
    print(_null_line(metric, n, f"rank {r} exited with code {code}" if failed else "rank 0 printed no JSON line",
                     stderr_tail=tail), flush=True)
    return 3
