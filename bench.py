#!/usr/bin/env python3
"""bench.py -- frames/s of the ORB extract + Hamming match hot path on MI355X.

Contract (see the task brief): `python bench.py --gpus N --steps K --warmup W`;
for N>1 launched by torch.distributed.run, one rank per GPU.  A *step* is one pass
of the hot path over one batch: 64 synthetic 640x480 frames per GPU ->
ORBextractor (2000 features, 8 levels, FAST 20/7) -> 2000x2000 brute-force
Hamming match of every frame against its successor in the batch (+ TH_LOW / 0.6
ratio filter) -> gather of the fixed-size result records on rank 0 (N>1 only).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

`roofline`: dominant kernel's algorithmic bytes / its mean launch time, timed
with HIP events on the launch stream inside the timed region.  `cpu_baseline`:
the CPU oracle (single thread) on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, BATCH = 640, 480, 64
PARAMS = (2000, 1.2, 8, 20, 7)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Integer VALU issue rate: one wave64 instruction per 4 clocks per SIMD (tools/ubench/valu.hip), 1024 SIMDs, 2.4 GHz
VALU_PEAK_GINST = 1024 * 2.4 / 4.0
# MI355X_MICROARCH.md, Matrix cores: FP4 (v_mfma_f32_32x32x64_f8f6f4) dense peak; the all-pairs matcher runs on it
MFMA_FP4_PEAK_TFLOPS = 10000.0
# SURVEY 8(d) algorithmic bytes per 640x480 frame, by stage
PYR_PX = 950532
ALG_BYTES = {
    "k_pyr_level0": 307200 + 307200,            # read L0 input + write level 0
    "k_pyr_resize": (PYR_PX - 307200) * 2,      # read level l-1 (approx. its own size class) + write level l, levels 1..7
    "k_fast_cells": PYR_PX,                     # FAST read of the pyramid
    "k_octree": 0,
    "k_blur": 2 * PYR_PX,                       # blur read + write
    "k_describe": 2000 * 749 + 2000 * 512 + 2000 * 60,
    "k_match_sets_mfma": 144000,                # 2 x 2000 x 32 B read + 2000 x 8 B written
}
FRAME_BYTES = 6751328  # whole extract path per frame


def cpu_baseline(min_seconds=12.0, ndistinct=16):
    """Oracle (CPU restatement, one thread) on a bounded sample of the workload
    (about 10-20 s of CPU work)."""
    import oracle
    from orb_slam2_e_amd.synth import synth_frame
    imgs = [synth_frame(k) for k in range(ndistinct)]
    o = oracle.OrbOracle(*PARAMS)
    o.extract(imgs[0])  # warm
    t0 = time.perf_counter()
    prev = None
    nframes = 0
    while time.perf_counter() - t0 < min_seconds:
        desc = o.extract(imgs[nframes % ndistinct])[1]
        if prev is not None:
            b, s, ix = oracle.match_bruteforce(prev, desc)
            oracle.match_filter(b, s, ix, 45, 0.6)
        prev = desc
        nframes += 1
    dt = time.perf_counter() - t0
    return {"value": nframes / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{nframes} synthetic 640x480 frames ({ndistinct} distinct): oracle extract + 2000x2000 match "
                      f"against the previous frame, 1 thread, {dt:.1f} s"}


def fem_bench(rank, world, dist, torch, dev, cdev, nmesh=256, iters=200, cpu=True):
    """Config 3 (10,368-tet / 6,591-dof mesh, E=3500, nu=0.495): assemble K + 200
    CG iterations, single mesh and a batch of `nmesh` distinct matrices per GPU."""
    from orb_slam2_e_amd.fem import FEA2, FEM_TET4
    from orb_slam2_e_amd.synth import synth_tet_batch

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    out = {}
    for label, nm in (("single", 1), ("batch", nmesh)):
        nodes, tets, fixed, load = synth_tet_batch(nm, 12, seed=11 + 7919 * rank)
        fea = FEA2(nodes, tets, FEM_TET4)
        fea.profile(True)
        t0 = time.perf_counter(); fea.MatrixAssembly(); t_asm = time.perf_counter() - t0
        fea.eliminate_dofs(fixed)
        b = np.tile(load, (nm, 1)); b[:, fixed] = 0
        fea.cg_setup(b)
        fea.profile(True)
        fea.cg_iterate(20); fea.cg_result()           # warm + per-kernel split (untimed pass, all kinds)
        split = {k: v[0] / max(v[1], 1) for k, v in fea.profile_read().items() if v[1]}
        fea.cg_setup(b)
        fea.profile(4 if nm > 1 else 0)               # timed region: events around k_fem_spmv only (batch);
        barrier()                                     # the single mesh replays a hipGraph (no events inside)
        t0 = time.perf_counter()
        fea.cg_iterate(iters)
        x, rel = fea.cg_result()                       # synchronises
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        prof = fea.profile_read()
        n, nnz = fea.Ksize, fea.nnz
        spmv_ms = prof["k_fem_spmv"][0] / max(prof["k_fem_spmv"][1], 1) if prof["k_fem_spmv"][1] else split.get("k_fem_spmv", 0.0)
        spmv_bytes = nm * (nnz * 8 + (n + 1) * 4 + 2 * n * 8)      # nnz(val+4)+(n+1)4+2n*val', f32 K, f64 x/y
        iter_bytes = spmv_bytes + nm * (2 * 2 + 3 * 3) * n * 8
        out[label] = {"meshes_per_gpu": nm, "n_dof": n, "nnz": nnz, "cg_iters": iters,
                      "cg_mesh_iters_per_s": world * nm * iters / dt, "ms_per_iter": dt / iters * 1e3,
                      "assemble_ms": t_asm * 1e3, "relres_after": float(rel.max()),
                      "spmv_avg_launch_ms": spmv_ms, "spmv_alg_bytes_per_launch": spmv_bytes,
                      "spmv_GBps": spmv_bytes / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0,
                      "cg_iter_GBps": iter_bytes * iters / dt / 1e9,
                      "kernel_ms_per_launch_untimed_pass": split}
        if label == "single" and cpu and rank == 0:
            import oracle
            rp, col, val = fea.csr()
            t0 = time.perf_counter(); reps = 0
            while time.perf_counter() - t0 < 10.0:
                oracle.fem_cg(rp, col, val, b[0], iters, 0.0); reps += 1
            dtc = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": reps * iters / dtc, "unit": "CG iters/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle Jacobi-PCG, {reps} x {iters} iterations on the single 6,591-dof mesh, {dtc:.1f} s"}
        del fea
    bt = out["batch"]
    out["roofline"] = {"bound": "hbm", "kernel": "k_fem_spmv", "achieved": bt["spmv_GBps"], "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": bt["spmv_GBps"] / HBM_PEAK_GBS, "traffic": None,
                       "avg_launch_ms": bt["spmv_avg_launch_ms"], "alg_bytes_per_launch": bt["spmv_alg_bytes_per_launch"]}
    tfile = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tfile) and nmesh == 256:      # PMC passes were taken on the 256-mesh batch
        tr = json.load(open(tfile)).get("k_fem_spmv")
        if tr:
            out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = tr["source"]
    return out


def matcher_loops_bench(cpu=True):
    """Tracking-time matching: one call = one whole ORBmatcher loop, host arrays in and out
    (PCIe and the per-call host work included), against the oracle's literal loop on the same inputs."""
    from orb_slam2_e_amd.matcher import ORBmatcher
    from orb_slam2_e_amd.synth import synth_bow_case, synth_projection_case
    from orb_slam2_e_amd.vocabulary import feature_vector_arrays

    def ms(f, reps):
        for _ in range(5): f()                      # the card idles (and clocks down) while the host times the CPU leg
        t0 = time.perf_counter()
        for _ in range(reps): f()
        return (time.perf_counter() - t0) / reps * 1e3

    q, qd, qa, takes, kps, desc, bounds, occ, ur = synth_projection_case(0, n=2000, nq=2000, hot=2000)
    d1, a1, node1, keep1, valid1, d2, a2, node2, keep2, valid2 = synth_bow_case(0)
    fv1 = feature_vector_arrays(node1, keep1); fv2 = feature_vector_arrays(node2, keep2)
    m = ORBmatcher(0.6, True)
    out = {"search_by_projection_2000x2000_ms": ms(lambda: m.search_projection(q, qd, qa, takes, kps, desc, bounds, occ, ur, 95), 50),
           "search_by_bow_2000x2100_ms": ms(lambda: m.SearchByBoW(fv1, valid1, d1, a1, fv2, valid2, d2, a2, False), 50),
           "search_window_2000x2000_ms": ms(lambda: m.search_window(q, qd, kps, desc, bounds, occ, ur), 50)}
    if cpu:
        import oracle
        ofv1 = oracle.feature_vector(node1, keep1); ofv2 = oracle.feature_vector(node2, keep2)
        out["cpu_baseline"] = {
            "kind": "port", "cores": 1, "unit": "ms per call", "sample": "the same inputs, 5 calls each",
            "search_by_projection_2000x2000_ms": ms(lambda: oracle.search_projection_seq(q, qd, qa, takes, kps, desc, bounds, occ, ur, 95, 0.6, False, True), 5),
            "search_by_bow_2000x2100_ms": ms(lambda: oracle.search_by_bow(ofv1, valid1, d1, a1, ofv2, None, d2, a2, False, 0.6, True), 5)}
    return out


def stereo_bench(cpu=True):
    """Config 5: one 1242x375 KITTI-shaped pair through the drop-in calls -- left and right ORBextractor::operator()
    (host image in, keypoints / descriptors out) and Frame::ComputeStereoMatches on the two resident pyramids."""
    from orb_slam2_e_amd import ComputeStereoMatches, ORBextractor
    from orb_slam2_e_amd.synth import synth_stereo_pair
    fx, bf = 718.856, 386.1448                      # Examples/Stereo/KITTI00-02.yaml:8,25
    mb = np.float32(bf) / np.float32(fx)
    left, right = synth_stereo_pair(0)
    eL, eR = ORBextractor(*PARAMS), ORBextractor(*PARAMS)

    def frame():
        eL(left); eR(right)
        return ComputeStereoMatches(eL, eR, mb, np.float32(bf))

    u, d = frame()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps): frame()
    t_all = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): ComputeStereoMatches(eL, eR, mb, np.float32(bf))
    t_st = (time.perf_counter() - t0) / reps
    out = {"pair": "1242x375, 2000 features per image", "stereo_frame_ms": t_all * 1e3, "compute_stereo_matches_ms": t_st * 1e3,
           "stereo_pairs_per_s": 1.0 / t_all, "matched": int((u >= 0).sum())}
    if cpu:
        import oracle
        oL, oR = oracle.OrbOracle(*PARAMS), oracle.OrbOracle(*PARAMS)
        t0 = time.perf_counter()
        kL, dL = oL.extract(left); kR, dR = oR.extract(right)
        t1 = time.perf_counter()
        oracle.stereo_matches(oL, oR, kL, dL, kR, dR, mb, np.float32(bf))
        t2 = time.perf_counter()
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "unit": "ms per pair", "sample": "the same pair, once",
                               "stereo_frame_ms": (t2 - t0) * 1e3, "compute_stereo_matches_ms": (t2 - t1) * 1e3}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="N > 1: steps of a context whose records travel in one RCCL gather (default: every step; larger values "
                         "= fewer, larger collectives, each gather costing two cross-stream dependencies; partial buckets are "
                         "flushed before every barrier).  Not measurable on the one-GPU development box: tune against SCALE_rNN.json")
    ap.add_argument("--pipeline", type=int, default=3, help="independent contexts/streams the steps rotate over")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs; gloo only to rehearse N>1 on one GPU")
    ap.add_argument("--host-io", action="store_true", help="also time the PCIe-inclusive path (host images in, host results out)")
    ap.add_argument("--no-fem", action="store_true")
    ap.add_argument("--match-kernel", choices=["auto", "popcount"], default="auto",
                    help="all-pairs matcher: auto = FP4 matrix-core kernel (default), popcount = XOR + v_bcnt kernel; same results")
    ap.add_argument("--fem-meshes", type=int, default=256)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rehearse = args.dist_backend != "nccl"   # all ranks on GPU 0, collectives staged through the CPU (gloo)
    dev_index = 0 if (world == 1 or rehearse) else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(args.dist_backend)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    dev = torch.device("cuda", dev_index)
    cdev = torch.device("cpu") if rehearse else dev   # where collective payloads live

    from orb_slam2_e_amd import ORBextractor, ORBmatcher
    from orb_slam2_e_amd._lib import lib
    from orb_slam2_e_amd.synth import synth_frames

    # ---- inputs resident in HBM (weak scaling: each rank owns its own 64 frames)
    frames = synth_frames(BATCH, W, H, start=rank * BATCH)
    d_frames = torch.from_numpy(frames).to(dev)
    m = ORBmatcher(0.6)
    L = lib()
    if args.match_kernel == "popcount":
        ORBmatcher.set_allpairs_kernel(ORBmatcher.ALLPAIRS_POPCOUNT)
    qa = torch.arange(BATCH, dtype=torch.int32, device=dev)
    qb = ((qa + 1) % BATCH).to(torch.int32)

    # `--pipeline P` independent contexts (extractor workspace + result buffers + HIP stream): step k runs
    # on context k % P, so the latency-bound stages of one batch (pyramid, octree) overlap the VALU-bound
    # stages of the next (FAST, match).  Every step still does the full work on its own buffers.
    class Ctx:
        pass
    ctxs = []
    for i in range(max(1, args.pipeline)):
        c = Ctx()
        c.ex = ORBextractor(*PARAMS)
        c.tstream = torch.cuda.Stream(device=dev) if args.pipeline > 1 else torch.cuda.current_stream()
        c.stream = c.tstream.cuda_stream
        cap = c.ex.capacity
        c.best = torch.empty((BATCH, cap), dtype=torch.int32, device=dev)
        c.second = torch.empty_like(c.best); c.idx = torch.empty_like(c.best); c.match12 = torch.empty_like(c.best)
        c.nmatch = torch.zeros(BATCH, dtype=torch.int32, device=dev)
        c.ex.extract_batch_device(d_frames.data_ptr(), BATCH, H, W, c.stream)  # sizes the workspace
        c.kps_p, c.desc_p, c.cnt_p, _ = c.ex.result_dev()
        ctxs.append(c)
    torch.cuda.synchronize()
    # fixed-size records {kps[CAP], desc[CAP][32], counts, match12[CAP], nmatch} -> one send buffer
    n_kps, n_desc, n_cnt, n_m12 = cap * 28 * BATCH, cap * 32 * BATCH, 4 * BATCH, cap * 4 * BATCH
    rec_bytes = n_kps + n_desc + n_cnt + n_m12 + n_cnt
    counts_t = torch.zeros(BATCH, dtype=torch.int32, device=dev)
    do_gather = world > 1 and not args.no_gather
    GE = max(1, args.gather_every)
    for c in ctxs:
        c.nfill = 0   # records waiting in c.send
        c.send = torch.empty(GE * rec_bytes, dtype=torch.uint8, device=dev) if do_gather else None
        c.recv = [torch.empty(GE * rec_bytes, dtype=torch.uint8, device=cdev) for _ in range(world)] if (do_gather and rank == 0) else None

    def pack_records(c):
        base = c.nfill * rec_bytes
        p0 = c.send.data_ptr() + base
        L.orbx_copy_results_dev(c.ex._h, C.c_void_p(p0), C.c_void_p(p0 + n_kps), C.c_void_p(p0 + n_kps + n_desc),
                                C.c_void_p(c.stream))
        o = base + n_kps + n_desc + n_cnt
        c.send[o:o + n_m12].copy_(c.match12.view(torch.uint8).reshape(-1), non_blocking=True)
        c.send[o + n_m12:o + n_m12 + n_cnt].copy_(c.nmatch.view(torch.uint8).reshape(-1), non_blocking=True)
        c.nfill += 1

    def gather_bucket(c):
        """The records of the last c.nfill steps of this context -> rank 0 (every rank holds the same number)."""
        n = c.nfill * rec_bytes
        if not do_gather or n == 0:
            return
        with torch.cuda.stream(c.tstream):
            recv = [r[:n] for r in c.recv] if rank == 0 else None
            dist.gather(c.send[:n].cpu() if rehearse else c.send[:n], recv, dst=0)
        c.nfill = 0

    def step(k):
        c = ctxs[k % len(ctxs)]
        with torch.cuda.stream(c.tstream):
            c.ex.extract_batch_device(d_frames.data_ptr(), BATCH, H, W, c.stream)
            m.match_batch_device(c.desc_p, c.cnt_p, cap, qa.data_ptr(), qb.data_ptr(), BATCH, c.best.data_ptr(),
                                 c.second.data_ptr(), c.idx.data_ptr(), c.match12.data_ptr(), c.nmatch.data_ptr(),
                                 stream=c.stream)
            if do_gather:
                pack_records(c)
                if c.nfill == GE:
                    gather_bucket(c)

    def sync():
        for c in ctxs:
            gather_bucket(c)   # partial buckets: every step's records are on rank 0 before the barrier
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()

    def read_profiles():
        kern = {}
        for c in ctxs:
            names = (C.c_char_p * 16)(); ms = (C.c_double * 16)(); ln = (C.c_int64 * 16)(); nk = C.c_int(0)
            L.orbx_profile_read(c.ex._h, 16, names, ms, ln, C.byref(nk))
            for i in range(nk.value):
                a, b = kern.get(names[i].decode(), (0.0, 0))
                kern[names[i].decode()] = (a + ms[i], b + ln[i])
        mm, ml = C.c_double(0), C.c_int64(0)
        L.orbm_profile_read(C.byref(mm), C.byref(ml))
        kern["k_match_sets_mfma"] = (mm.value, ml.value)
        return kern

    # untimed pass with events around every kernel: the per-kernel split and the dominant kernel
    KINDS = ["k_pyr_level0", "k_pyr_resize", "k_fast_cells", "k_octree", "k_blur", "k_describe"]
    for c in ctxs:
        L.orbx_profile_enable(c.ex._h, -1)
    L.orbm_profile_enable(1)
    nprof = 3
    for k in range(nprof):      # one context, one step at a time: no overlap in the per-kernel split
        step(0)
        torch.cuda.synchronize()
    sync()
    kern_all = read_profiles()
    split_ms = {k: v[0] / nprof for k, v in kern_all.items()}
    # a start/stop event pair inflates each measured launch by ~10-20 us (rocprof traces in profiles/); rank the
    # kinds by time net of 15 us per launch so that the 7-launch resize chain is not picked for its event overhead
    dom = max(split_ms, key=lambda k: split_ms[k] - 0.015 * kern_all[k][1] / nprof)
    # timed region: events around the dominant kernel only (each event costs dispatch-gap time)
    for c in ctxs:
        L.orbx_profile_enable(c.ex._h, (1 << KINDS.index(dom)) if dom in KINDS else 0)
    L.orbm_profile_enable(1 if dom == "k_match_sets_mfma" else 0)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- dominant kernel's launch times (HIP events on the launch stream, recorded in the timed region)
    kern = read_profiles()
    for c in ctxs:
        L.orbx_profile_enable(c.ex._h, 0)
    L.orbm_profile_enable(0)
    ex, nmatch, stream = ctxs[0].ex, ctxs[0].nmatch, ctxs[0].stream

    host_io = None
    if args.host_io and rank == 0:
        # PCIe-inclusive rate: 64 host frames in, every frame's keypoints / descriptors / counts out.  (a) the plain
        # synchronous calls on pageable memory; (b) what a batch front-end does: pinned buffers, the upload, the kernels
        # and the download of a batch queued on one stream per context, contexts rotating.
        c = ctxs[0]
        c.ex.extract_batch(frames); c.ex.download_batch()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            c.ex.extract_batch(frames)
            c.ex.download_batch()
        rate_sync = reps * BATCH / (time.perf_counter() - t0)
        for c in ctxs:
            c.pin_in = torch.from_numpy(frames).pin_memory()
            c.d_in = torch.empty((BATCH, H, W), dtype=torch.uint8, device=dev)
            c.pin_kps = torch.empty(cap * 28 * BATCH, dtype=torch.uint8).pin_memory()
            c.pin_desc = torch.empty(cap * 32 * BATCH, dtype=torch.uint8).pin_memory()
            c.pin_cnt = torch.empty(BATCH, dtype=torch.int32).pin_memory()

        def io_step(k):
            c = ctxs[k % len(ctxs)]
            with torch.cuda.stream(c.tstream):
                c.d_in.copy_(c.pin_in, non_blocking=True)
                c.ex.extract_batch_device(c.d_in.data_ptr(), BATCH, H, W, c.stream)
                L.orbx_copy_results_dev(c.ex._h, C.c_void_p(c.pin_kps.data_ptr()), C.c_void_p(c.pin_desc.data_ptr()),
                                        C.c_void_p(c.pin_cnt.data_ptr()), C.c_void_p(c.stream))
        for k in range(30): io_step(k)
        torch.cuda.synchronize()
        reps = 300
        t0 = time.perf_counter()
        for k in range(reps): io_step(k)
        torch.cuda.synchronize()
        rate_pipe = reps * BATCH / (time.perf_counter() - t0)
        big_h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory(); big_d = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        bw = {}
        for name, dst, src in (("h2d", big_d, big_h), ("d2h", big_h, big_d)):
            dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            bw[name] = 5 * (64 << 20) / (time.perf_counter() - t0) / 1e9
        host_io = {"pcie_pinned_copy_GBps": bw, "frames_per_s_extract_only_sync_pageable": rate_sync, "frames_per_s_extract_only_pinned_pipelined": rate_pipe,
                   "bytes_per_frame": W * H + cap * 60 + 4,
                   "note": "64 host u8 frames in over PCIe, all keypoints+descriptors+counts out; no match"}

    fem = None
    if not args.no_fem:
        fem = fem_bench(rank, world, dist, torch, dev, cdev, nmesh=args.fem_meshes, cpu=not args.no_cpu_baseline)

    if rank == 0:
        total_frames = world * BATCH * args.steps
        value = total_frames / dt
        launches_per_step = max(kern[dom][1] // max(args.steps, 1), 1)
        avg_launch_ms = kern[dom][0] / max(kern[dom][1], 1)
        alg = ALG_BYTES[dom] * BATCH / launches_per_step  # algorithmic bytes per launch
        achieved = alg / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        L.orbx_copy_results_dev(ex._h, None, None, C.c_void_p(counts_t.data_ptr()), C.c_void_p(stream))
        torch.cuda.synchronize()
        counts = counts_t
        out = {
            "metric": "frames/s ORB extract+match (640x480, 2000 feat)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"{world}xMI355X: {world * BATCH}-frame batch sharded {BATCH}/GPU, " if world > 1 else
                                    f"1xMI355X: batch of {BATCH} ") + "synthetic 640x480 frames, ORB extract (2000 feat, 8 levels) "
                                   "+ 2000x2000 brute-force Hamming match per frame" + (", RCCL gather on rank 0" if world > 1 else ""),
                       "frames_per_gpu": BATCH, "global_batch": world * BATCH, "pipeline_contexts": len(ctxs),
                       "allpairs_kernel": "k_match_sets (popcount)" if args.match_kernel == "popcount" else "k_match_sets_mfma (FP4 matrix cores)",
                       "parallelism": f"frames sharded {BATCH}/GPU, results gathered on rank 0 ({GE} steps per RCCL gather)" if world > 1 else "single GPU",
                       "mean_keypoints_per_frame": float(counts.float().mean().item()),
                       "mean_matches_per_frame": float(nmatch.float().mean().item())},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_ms": avg_launch_ms, "alg_bytes_per_launch": alg},
            "pipeline_hbm_frac": value / world * FRAME_BYTES / 1e9 / HBM_PEAK_GBS,
            "kernel_ms_per_step_untimed_pass": split_ms,
        }
        # the all-pairs matcher is the one contraction on the path: 2 x 256 FP4 multiply-adds per descriptor pair
        nk = counts.double().clamp(max=cap).cpu()
        pair_flops = 512.0 * float((nk[qa.cpu().long()] * nk[qb.cpu().long()]).sum())
        mm_ms = split_ms.get("k_match_sets_mfma", 0.0)
        if mm_ms > 0 and args.match_kernel != "popcount":
            out["matcher_mfma"] = {"kernel": "k_match_sets_mfma", "bound": "mfma", "achieved": pair_flops / (mm_ms * 1e-3) / 1e12,
                                   "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": pair_flops / (mm_ms * 1e-3) / 1e12 / MFMA_FP4_PEAK_TFLOPS,
                                   "flops_per_launch": pair_flops, "launch_ms_untimed_pass": mm_ms,
                                   "note": "FP4 MFMA computes the selection keys; the top-2 fold (2 VALU instructions per pair) bounds it"}
        if dom == "k_match_sets_mfma" and args.match_kernel != "popcount":
            tf = pair_flops / (avg_launch_ms * 1e-3) / 1e12
            out["roofline"].update({"bound": "mfma", "achieved": tf, "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": tf / MFMA_FP4_PEAK_TFLOPS, "flops_per_launch": pair_flops})
        tfile = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tfile):   # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE)
            tr = json.load(open(tfile))
            if dom in tr:
                out["roofline"]["traffic"] = tr[dom]["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = tr[dom]["source"]
                if tr[dom].get("valu_wave_insts_per_launch"):
                    # the kernels that dominate this path are bound by integer VALU issue, not by HBM:
                    # the same launch time priced against the instruction count of the committed PMC pass
                    ginst = tr[dom]["valu_wave_insts_per_launch"] / (avg_launch_ms * 1e-3) / 1e9
                    out["roofline"]["valu"] = {"achieved": ginst, "peak": VALU_PEAK_GINST, "unit": "G wave64-instr/s",
                                               "frac": ginst / VALU_PEAK_GINST,
                                               "wave_insts_per_launch": tr[dom]["valu_wave_insts_per_launch"]}
        if host_io is not None:
            out["host_io"] = host_io
        if fem is not None:
            out["fem"] = fem
        if not args.no_fem:
            out["matcher_loops"] = matcher_loops_bench(cpu=not args.no_cpu_baseline)
            out["stereo"] = stereo_bench(cpu=not args.no_cpu_baseline)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
