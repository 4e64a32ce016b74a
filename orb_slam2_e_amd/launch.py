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


# ---- one rank, one set of cores ---------------------------------------------------------------------------------------------
# A rank's step is ~17 kernel launches per 0.26 ms from one host thread; eight ranks left to the scheduler migrate across sockets
# and share cores with each other's HIP helper threads.  Each rank pins itself -- before it touches the GPU, so that the runtime's
# threads inherit the mask -- to its share of the CPUs local to ITS GPU (PCI locality from sysfs), or to a contiguous share of the
# CPUs it may use when the topology cannot be read.

def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def _drm_local_cpus(index, sysfs):
    """The AMD render nodes' PCI functions in bus order (the order the runtime enumerates the GPUs of one node in), the index-th one's
    local_cpulist.  Readable without privileges, unlike the KFD topology's per-node properties."""
    drm = os.path.join(sysfs, "class/drm")
    devs = set()
    for name in os.listdir(drm):
        if not name.startswith("renderD"):
            continue
        dev = os.path.realpath(os.path.join(drm, name, "device"))
        try:
            if open(os.path.join(dev, "vendor")).read().strip().lower() != "0x1002":
                continue
        except OSError:
            continue
        devs.add(dev)
    dev = sorted(devs, key=os.path.basename)[index]
    return _parse_cpulist(open(os.path.join(dev, "local_cpulist")).read()) or None


def gpu_local_cpus(index, sysfs="/sys"):
    """CPUs on the NUMA node of HIP device `index` (KFD topology order; else the render nodes in PCI bus order), or None when it
    cannot be told."""
    try:
        return _kfd_local_cpus(index, sysfs)
    except Exception:
        pass
    try:
        return _drm_local_cpus(index, sysfs)
    except Exception:
        return None


def _kfd_local_cpus(index, sysfs):
    if True:
        nodes = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
        gpus = []
        for n in sorted(os.listdir(nodes), key=int):
            props = dict(line.split()[:2] for line in open(os.path.join(nodes, n, "properties")) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        p = gpus[index]
        loc, dom = int(p["location_id"]), int(p.get("domain", "0"))
        bdf = "%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        return _parse_cpulist(open(os.path.join(sysfs, "bus/pci/devices", bdf, "local_cpulist")).read()) or None


def rank_cpu_mask(local_rank, world, avail, local_cpus_of=None):
    """CPUs for rank `local_rank` of `world` on this node.  avail: CPUs this process may use; local_cpus_of(r): CPUs near rank r's
    GPU or None.  Ranks whose GPUs share a locality split it evenly, in rank order; without locality: a contiguous split of avail.
    Every rank gets at least one CPU; masks of different ranks are disjoint whenever there are at least `world` CPUs."""
    avail = sorted(avail)
    if world <= 1 or not avail:
        return avail
    near = [local_cpus_of(r) if local_cpus_of else None for r in range(world)]
    if all(n for n in near):
        mine = sorted(set(near[local_rank]) & set(avail))
        peers = [r for r in range(world) if sorted(set(near[r]) & set(avail)) == mine]
        if mine and len(mine) >= len(peers):
            per = len(mine) // len(peers)
            i = peers.index(local_rank)
            return mine[i * per:(i + 1) * per]
    per = max(1, len(avail) // world)
    lo = (local_rank * per) % len(avail)
    return avail[lo:lo + per]


def visible_device_index(index, env=None):
    """Physical GPU behind HIP device `index` of this process: ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES is
    an alias on ROCm) renumber the devices a process sees; a list of plain indices is mapped through, anything else (UUIDs, an
    index beyond the list) gives None = unknown."""
    env = os.environ if env is None else env
    phys = index
    for name in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):    # HIP's filter applies on top of ROCr's
        v = env.get(name)
        if v is None or v.strip() == "":
            continue
        items = [x.strip() for x in v.split(",")]
        if not all(x.isdigit() for x in items) or phys >= len(items):
            return None
        phys = int(items[phys])
        if name != "ROCR_VISIBLE_DEVICES" and env.get("ROCR_VISIBLE_DEVICES") is None:
            break
    return phys


def pin_rank(local_rank, world, dev_index=None):
    """Pins the calling process (call before the first GPU call); returns what it did, for the bench line.  dev_index: the HIP
    device this rank will use (default: local_rank).  Ranks that map one to one onto devices get the CPUs local to the PHYSICAL GPU
    behind their device (visible-device lists mapped through); when ranks share a device (a gloo rehearsal: all on GPU 0) or the
    physical GPU cannot be told, every rank gets a contiguous share of the CPUs instead."""
    if world <= 1:
        return None
    try:
        avail = sorted(os.sched_getaffinity(0))
        dev_index = local_rank if dev_index is None else dev_index
        local = None
        if dev_index == local_rank:                                  # rank r <-> device r

            def local(r):
                p = visible_device_index(r)
                return gpu_local_cpus(p) if p is not None else None
        mask = rank_cpu_mask(local_rank, world, avail, local)
        os.sched_setaffinity(0, mask)
        numa = local is not None and all(local(r) for r in range(world))      # (rank_cpu_mask uses the locality only if every rank has one)
        return {"cpus": "%d-%d" % (mask[0], mask[-1]) if mask == list(range(mask[0], mask[-1] + 1)) else ",".join(map(str, mask)),
                "n": len(mask), "from": "GPU-local CPUs (sysfs)" if numa else "contiguous share",
                "gpu": {"hip_device": dev_index, "physical": visible_device_index(dev_index)}}
    except Exception as e:       # never fatal: an unpinned rank is slower, not wrong
        return {"error": str(e)}
