"""`bench.py --gpus N` without a launcher around it: the parent starts N rank processes, relays rank 0's line, and
fails if any rank fails (one process per GPU; BASELINE config 4).

Imports the standard library only and must stay that way: the parent may not touch the GPU (a process that has
initialised HIP must not be replaced, and a parent holding the card would be a seventh user of it), so everything
here runs before `import torch`.  The ranks are fresh child processes (`subprocess`), never an exec of this one.
"""
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env(rank, world, port, base=None):
    """Environment of rank `rank` of `world` on this node: what torch.distributed.run would have set."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC only on this host driver (RCCL needs it)
    return env


def should_spawn(gpus, environ=None):
    """True when this process is the parent of an N > 1 run: asked for N > 1 and not already a rank of a launcher."""
    environ = os.environ if environ is None else environ
    return gpus > 1 and "WORLD_SIZE" not in environ


def check_world(gpus, environ=None):
    """A rank launched by torch.distributed.run (or by spawn_ranks) must agree with --gpus; raises SystemExit otherwise."""
    environ = os.environ if environ is None else environ
    world = int(environ.get("WORLD_SIZE", "1"))
    if world != gpus:
        raise SystemExit(f"bench: --gpus {gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {gpus} "
                         f"(or unset WORLD_SIZE and let bench.py start the ranks itself)")
    return world


def _pump(stream, sink, prefix, keep):
    for line in iter(stream.readline, ""):
        if keep is not None:
            keep.append(line)
        else:
            sink.write(prefix + line)
            sink.flush()
    stream.close()


def spawn_ranks(cmd, world, timeout=None, env=None, out=None, err=None, poll=0.2):
    """Start `cmd` (argv list) as ranks 0..world-1, wait for all of them.

    Rank 0's stdout is collected and written to `out` when it ends (its JSON line is the run's result); the other
    ranks' stdout and every rank's stderr go to `err` line by line, prefixed with the rank.  If a rank exits non-zero
    (or the timeout passes) the remaining ranks -- exactly the process groups started here -- are terminated and the
    first failing code is returned.  Returns (exit code, rank 0's stdout lines)."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port()
    procs, pumps, rank0_lines = [], [], []
    for r in range(world):
        p = subprocess.Popen(cmd, env=rank_env(r, world, port, env), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, bufsize=1, start_new_session=True)
        procs.append(p)
        for stream, keep in ((p.stdout, rank0_lines if r == 0 else None), (p.stderr, None)):
            t = threading.Thread(target=_pump, args=(stream, err, f"[rank {r}] ", keep), daemon=True)
            t.start()
            pumps.append(t)
    t0 = time.monotonic()
    code = 0
    try:
        while True:
            states = [p.poll() for p in procs]
            bad = [(r, s) for r, s in enumerate(states) if s not in (None, 0)]
            if bad:
                code = bad[0][1] if bad[0][1] > 0 else 128 - bad[0][1]
                err.write(f"bench: rank {bad[0][0]} exited with {bad[0][1]}; stopping the other ranks\n")
                break
            if all(s == 0 for s in states):
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                code = 124
                err.write(f"bench: ranks still running after {timeout} s; stopping them\n")
                break
            time.sleep(poll)
    finally:
        for p in procs:                       # only what was started here, by process group id
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
        deadline = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                p.wait()
        for t in pumps:
            t.join(5)
    for line in rank0_lines:
        out.write(line)
    out.flush()
    return code, rank0_lines


def result_line(lines):
    """The last line of rank 0's stdout that parses as a JSON object, or None."""
    for line in reversed(lines):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def run_parent(script, argv, gpus, timeout=None, env=None, out=None, err=None):
    """What `python bench.py --gpus N ...` does when it is not yet a rank: N ranks of the same command line, rank 0's
    JSON line relayed, exit code non-zero if a rank failed or the line does not report n_gpus == N."""
    err = sys.stderr if err is None else err
    code, lines = spawn_ranks([sys.executable, script] + list(argv), gpus, timeout=timeout, env=env, out=out, err=err)
    if code != 0:
        return code
    res = result_line(lines)
    if res is None:
        err.write("bench: rank 0 printed no result line\n")
        return 3
    if res.get("n_gpus") != gpus:
        err.write(f"bench: asked for --gpus {gpus}, the result line says n_gpus = {res.get('n_gpus')}\n")
        return 4
    return 0
