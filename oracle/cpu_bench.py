#!/usr/bin/env python3
"""CPU baseline of bench.py (SURVEY 8d protocol).  TEST / MEASUREMENT INFRASTRUCTURE, never part of the product.

Runs the oracle (kind "port": the CPU restatement of the reference's algorithm, built -O3 -march=native
-ffp-contract=off on THIS machine as liboracle_native.so) on a bounded sample of the bench workload:
  * single thread pinned to one core (sched_setaffinity = taskset -c), 3 warm-ups, median of >= 20 runs of
    extract (one 640x480 frame), match (2000x2000 Hamming + best/second + TH_LOW / ratio filter) and the
    frames/s of extract + match against the previous frame;
  * all host cores: one worker process per core, each pinned, one frame per worker at a time.
Prints one JSON object.  Started by bench.py as a child process (it never touches the GPU).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

os.environ["ORACLE_NATIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PARAMS = (2000, 1.2, 8, 20, 7)


def _pin(cpu):
    try:
        os.sched_setaffinity(0, {cpu})
        return True
    except OSError:
        return False


def _usable_cores(cpus):
    """Cores this process may really use at once: its affinity mask, cut down to the cgroup CPU quota where one is
    readable, else to 16 (a one-GPU box hands out one GPU's share of the host, 16 cores, whatever the mask says);
    ORBX_CPU_WORKERS overrides.  Returns (n, how)."""
    n = len(cpus)
    if os.environ.get("ORBX_CPU_WORKERS"):
        return min(n, max(1, int(os.environ["ORBX_CPU_WORKERS"]))), "ORBX_CPU_WORKERS"
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / 100000.0)):
        try:
            q = parse(open(path).read())
            if q:
                return min(n, max(1, int(q))), "cgroup CPU quota"
        except (OSError, ValueError, IndexError):
            pass
    return (n, "affinity mask") if n <= 16 else (16, "affinity mask has %d cores, no cgroup quota readable: one GPU's share of 16 used" % n)


def _extract_match(o, oracle, imgs, prev):
    kps, desc = o.extract(imgs)
    if prev is not None:
        b, s, ix = oracle.match_bruteforce(prev, desc)
        oracle.match_filter(b, s, ix, 45, 0.6)
    return desc


def _worker(cpu, frames, reps, q, barrier):
    import oracle
    _pin(cpu)
    o = oracle.OrbOracle(*PARAMS)
    prev = _extract_match(o, oracle, frames[0], None)
    barrier.wait()
    t0 = time.perf_counter()
    n = 0
    for r in range(reps):
        for f in frames:
            prev = _extract_match(o, oracle, f, prev)
            n += 1
    q.put((n, t0, time.perf_counter()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=512)   # ~11 s of one core at 22 ms per frame (the brief: 10-30 s of CPU work)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--all-cores-frames", type=int, default=16, help="frames per worker in the all-cores pass")
    ap.add_argument("--legs", default="extract", help="comma list of extract, fem, stereo, loops, bow")
    ap.add_argument("--fem-csr", default=None, help=".npz with rp, col, val, b of the single config-3 mesh (written by bench.py)")
    ap.add_argument("--fem-iters", type=int, default=200)
    args = ap.parse_args()
    legs = args.legs.split(",")
    result = {}
    if "extract" in legs:
        result["extract_match"] = extract_leg(args)
    if "fem" in legs and args.fem_csr:
        result["fem"] = fem_leg(args)
    if "stereo" in legs:
        result["stereo"] = stereo_leg(args)
    if "loops" in legs:
        result["matcher_loops"] = loops_leg(args)
    if "bow" in legs:
        result["bow_transform"] = bow_leg(args)
    print(json.dumps(result), flush=True)


def _median_ms(f, runs=20, warmup=3):
    for _ in range(warmup): f()
    t = []
    for _ in range(runs):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return float(np.median(t)) * 1e3


PROTOCOL = "one thread pinned to one core, 3 warm-ups, median of 20 runs, -O3 -march=native -ffp-contract=off"


def fem_leg(args):
    import oracle
    _pin(sorted(os.sched_getaffinity(0))[0])
    z = np.load(args.fem_csr)
    rp, col, val, b = z["rp"], z["col"], z["val"], z["b"]
    ms = _median_ms(lambda: oracle.fem_cg(rp, col, val, b, args.fem_iters, 0.0))
    return {"value": args.fem_iters / (ms * 1e-3), "unit": "CG iters/s", "cores": 1, "kind": "port",
            "sample": f"oracle Jacobi-PCG, {args.fem_iters} iterations on the single {len(b)}-dof mesh (nnz {len(val)}); " + PROTOCOL}


def stereo_leg(args):
    import oracle
    from orb_slam2_e_amd.synth import synth_stereo_pair
    _pin(sorted(os.sched_getaffinity(0))[0])
    fx, bf = 718.856, 386.1448
    mb = np.float32(bf) / np.float32(fx)
    left, right = synth_stereo_pair(0)
    oL, oR = oracle.OrbOracle(*PARAMS), oracle.OrbOracle(*PARAMS)
    st = {}

    def frame():
        st["kL"], st["dL"] = oL.extract(left); st["kR"], st["dR"] = oR.extract(right)
        oracle.stereo_matches(oL, oR, st["kL"], st["dL"], st["kR"], st["dR"], mb, np.float32(bf))
    all_ms = _median_ms(frame)
    sm_ms = _median_ms(lambda: oracle.stereo_matches(oL, oR, st["kL"], st["dL"], st["kR"], st["dR"], mb, np.float32(bf)))
    return {"kind": "port", "cores": 1, "unit": "ms per pair", "sample": "the same 1242x375 pair; " + PROTOCOL,
            "stereo_frame_ms": all_ms, "compute_stereo_matches_ms": sm_ms}


def bow_leg(args):
    """The DBoW2 descent of bench.py's bow_transform leg (same synthetic k = 10, L = 6 tree, the first 10 of its 64 descriptor sets:
    20,000 descents per run) and the distinctive-descriptor call on the same 2000 map points."""
    import oracle
    sys.path.insert(0, ROOT)
    from bench import bow_case, distinctive_case
    _pin(sorted(os.sched_getaffinity(0))[0])
    voc, feats = bow_case()
    f = feats[:10].reshape(-1, 32)
    ms = _median_ms(lambda: oracle.bow_descend(*voc, f, 4))
    desc, off = distinctive_case()
    dms = _median_ms(lambda: oracle.distinctive_descriptors(desc, off))
    return {"value": len(f) / (ms * 1e-3), "unit": "descriptors/s", "cores": 1, "kind": "port",
            "sample": f"oracle_bow_transform on {len(f)} descriptors (10 of the leg's 64 sets), the same tree; " + PROTOCOL,
            "ms_per_2000_descriptors": ms * 2000 / len(f), "distinctive_descriptors_call_ms": dms}


def _whole_last_ms(oracle):
    """SearchByProjection(CurrentFrame, LastFrame, th, bMono) as the literal loop, projection included (grid build inside, as the
    reference's Frame constructor pays it once per frame -- timed apart below would flatter the CPU)."""
    from orb_slam2_e_amd.synth import synth_tracking_scene
    s = synth_tracking_scene(11)
    lm = s["last_mp"]
    return _median_ms(lambda: oracle.search_by_projection_last(s["kps"], s["desc"], None, s["occupied"], s["bounds"], s["cam"], s["mb"], s["mbf"],
                                                               s["Tcw"], s["scale_factors"], s["Tlw"], s["last_valid"], s["pos"][lm],
                                                               s["mp_desc"][lm], s["last_takes"], s["last_octave"], s["last_angle"], 7.0, True))


def loops_leg(args):
    import oracle
    from orb_slam2_e_amd.synth import synth_bow_case, synth_initialization_case, synth_projection_case
    _pin(sorted(os.sched_getaffinity(0))[0])
    q, qd, qa, takes, kps, desc, bounds, occ, ur = synth_projection_case(0, n=2000, nq=2000, hot=2000)
    d1, a1, node1, keep1, valid1, d2, a2, node2, keep2, valid2 = synth_bow_case(0)
    ofv1 = oracle.feature_vector(node1, keep1); ofv2 = oracle.feature_vector(node2, keep2)
    ik1, id1, ik2, id2, iprev, ibounds = synth_initialization_case(0)
    return {"search_for_initialization_2000x2200_ms": _median_ms(lambda: oracle.search_for_initialization(ik1, id1, ik2, id2, iprev, ibounds, 100, 0.9, True)),
            "kind": "port", "cores": 1, "unit": "ms per call", "sample": "the same inputs; " + PROTOCOL,
            "search_by_projection_2000x2000_ms": _median_ms(lambda: oracle.search_projection_seq(q, qd, qa, takes, kps, desc, bounds, occ, ur, 95, 0.6, False, True)),
            "search_by_bow_2000x2100_ms": _median_ms(lambda: oracle.search_by_bow(ofv1, valid1, d1, a1, ofv2, None, d2, a2, False, 0.6, True)),
            "search_by_projection_last_frame_whole_2000x2000_ms": _whole_last_ms(oracle)}


def extract_leg(args):
    import oracle
    from orb_slam2_e_amd.synth import synth_sequence
    cpus = sorted(os.sched_getaffinity(0))
    pinned = _pin(cpus[0])
    nd = min(max(args.runs, 2), 64)
    frames = synth_sequence(nd)
    o = oracle.OrbOracle(*PARAMS)
    descs = [None] * nd
    for i in range(args.warmup):
        descs[i % nd] = o.extract(frames[i % nd])[1]
    t_ex, t_m = [], []
    prev = descs[(args.warmup - 1) % nd] if args.warmup else None
    for i in range(args.runs):
        t0 = time.perf_counter()
        d = o.extract(frames[i % nd])[1]
        t1 = time.perf_counter()
        if prev is not None:
            b, s, ix = oracle.match_bruteforce(prev, d)
            oracle.match_filter(b, s, ix, 45, 0.6)
            t_m.append(time.perf_counter() - t1)
        t_ex.append(t1 - t0)
        prev = d
    ex_ms, m_ms = float(np.median(t_ex)) * 1e3, float(np.median(t_m)) * 1e3
    out = {"value": 1e3 / (ex_ms + m_ms), "unit": "frames/s", "cores": 1, "kind": "port",
           "extract_ms_median": ex_ms, "match_ms_median": m_ms, "runs": args.runs, "warmup": args.warmup,
           "pinned_to_cpu": cpus[0] if pinned else None,
           "flags": "-O3 -march=native -ffp-contract=off (oracle/Makefile: liboracle_native.so, built on this host)",
           "sample": f"{args.runs} frames of the bench sequence (640x480, 2000 features, 8 levels), one thread pinned to one core, "
                     f"{args.warmup} warm-ups; value = 1 / (median extract + median 2000x2000 match + filter)"}
    # all host cores: one pinned worker per core, one frame at a time per worker
    if pinned:
        os.sched_setaffinity(0, set(cpus))
    nproc, how = _usable_cores(cpus)
    cpus = cpus[:nproc]
    ctx = mp.get_context("fork")
    q, barrier = ctx.Queue(), ctx.Barrier(nproc)
    per = max(1, args.all_cores_frames)
    procs = [ctx.Process(target=_worker, args=(cpus[w], frames[(w * per + np.arange(per)) % nd], 1, q, barrier)) for w in range(nproc)]
    for p in procs: p.start()
    res = [q.get() for _ in procs]
    for p in procs: p.join()
    total = sum(r[0] for r in res)
    wall = max(r[2] for r in res) - min(r[1] for r in res)
    out["all_cores"] = {"value": total / wall, "unit": "frames/s", "cores": nproc, "nproc": os.cpu_count(),
                        "sample": f"{total} frames, one pinned worker process per usable core ({nproc}, from: {how}; "
                                  f"os.cpu_count() = {os.cpu_count()}), extract + match each, {wall:.2f} s wall"}
    return out


if __name__ == "__main__":
    main()
