"""Stress: every family of the C-ABI from several host threads at once, for a while -- extractor handles (one per thread), the
re-entrant matcher calls (workspace pool), stereo pairs, FEM models created / assembled / solved / destroyed per iteration (block
and stream caches) -- each result compared with what the same call returned single-threaded.  usage: stress_threads.py [seconds] [threads]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from orb_slam2_e_amd import ComputeStereoMatches, ORBextractor, ORBmatcher
from orb_slam2_e_amd.extractor import extract_pair
from orb_slam2_e_amd.fem import FEA2, FEM_C3D6, FEM_TET4, extrude_elems, second_layer
from orb_slam2_e_amd.synth import synth_bow_case, synth_frame, synth_initialization_case, synth_projection_case, synth_stereo_pair, synth_tet_mesh
from orb_slam2_e_amd.vocabulary import feature_vector_arrays

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
P = (1000, 1.2, 8, 20, 7)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def eq(a, b):
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(eq(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()
    return a == b


def make_jobs(k):
    """Closures returning comparable results; each thread owns its inputs (and its extractor handles)."""
    rng = np.random.default_rng(k)
    img = synth_frame(100 + k, 640, 480)
    left, right = synth_stereo_pair(k, w=800, h=300)
    ex, eL, eR = ORBextractor(*P), ORBextractor(*P), ORBextractor(*P)
    c = synth_projection_case(k, n=1500, nq=1500, hot=300)
    d1, a1, node1, keep1, valid1, d2, a2, node2, keep2, valid2 = synth_bow_case(k, n1=1200, n2=1300, nnodes=80)
    fv1, fv2 = feature_vector_arrays(node1, keep1), feature_vector_arrays(node2, keep2)
    ik1, id1, ik2, id2, iprev, ib = synth_initialization_case(k, n1=1200, n2=1300)
    m = np.load(os.path.join(GOLD, "fem_mesh_median.npz"))
    top, tris = m["points"], m["triangles"]
    p = top[tris]
    tris = tris[~((p[:, 0] == p[:, 1]).all(1) | (p[:, 0] == p[:, 2]).all(1) | (p[:, 1] == p[:, 2]).all(1))]
    nodes = second_layer(top, 0.5); elems = extrude_elems(tris, len(top))
    ids = np.arange(len(top), 2 * len(top), dtype=np.int32)
    pts = top.astype(np.float64) + 0.002 * (k + 1)
    tn, tt, tfixed, tload = synth_tet_mesh(5, 11 + k)
    tb = tload.copy(); tb[tfixed] = 0
    mb = np.float32(386.1448) / np.float32(718.856)

    def stereo():
        eL(left); eR(right)
        return ComputeStereoMatches(eL, eR, mb, np.float32(386.1448))

    pL, pR = ORBextractor(*P), ORBextractor(*P)      # first pair = workspace build, second = hipGraph capture, then replays: ALL of it
                                                     # inside the threads (nothing is captured before they start)

    def pair():
        return extract_pair(pL, pR, left, right)

    def fem_lm():
        fea = FEA2(nodes, elems, FEM_C3D6); fea.MatrixAssembly(); fea.ImposeDirichletEncastre_K(ids)
        fea.trial_setup(nodes.ravel(), ids, len(top), None)
        return fea.trial_energy(pts)

    def fem_cg():
        fea = FEA2(tn, tt, FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(tfixed)
        return fea.solve_cg(tb[None], iters=40)

    def fem_cg_two_level():
        fea = FEA2(tn, tt, FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(tfixed)
        fea.cg_preconditioner("two_level")
        x = fea.solve_cg(tb[None], iters=40)
        fea.cg_setup(tb[None]); fea.cg_iterate(50)          # the launch-per-phase path, under other threads' copies
        return x + fea.cg_result()

    return [("extract", lambda: ex(img)), ("stereo", stereo),
            ("projection", lambda: ORBmatcher(0.6, True).search_projection(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], 95)),
            ("window", lambda: ORBmatcher().search_window(c[0], c[1], c[4], c[5], c[6], c[7], c[8])),
            ("bruteforce", lambda: ORBmatcher().match_bruteforce(c[1][:700], c[5])),
            ("bow", lambda: ORBmatcher(0.7, True).SearchByBoW(fv1, valid1, d1, a1, fv2, valid2, d2, a2, False)),
            ("initialization", lambda: ORBmatcher(0.9, True).SearchForInitialization(ik1, id1, ik2, id2, iprev, ib, 100)),
            ("fem LM trial", fem_lm), ("fem CG", fem_cg), ("fem CG two-level", fem_cg_two_level), ("pair", pair)]


def collision_jobs():
    """The collision the round-4 run found: while ONE thread captures a hipGraph (the second orbx_extract_pair of a frame size), a
    blocking copy on the legacy stream from any other thread fails (hipErrorStreamCaptureImplicit) and kills the capture.  Every
    one-time set-up of the library that used to make such copies runs here on other threads WHILE thread 0 captures over and over
    (fresh handles each time: build, capture, replay): first-frame workspace builds, vocabulary uploads, first stereo calls, frame
    layouts.  Each job returns a comparable result; any error return raises in the Python mirror."""
    from orb_slam2_e_amd.matcher import Frame
    from orb_slam2_e_amd.vocabulary import ORBVocabulary
    from orb_slam2_e_amd.synth import synth_vocabulary
    sizes = [(320, 240), (400, 296), (352, 288), (480, 272)]
    imgs = {wh: (synth_frame(7, *wh), synth_frame(8, *wh)) for wh in sizes}
    voc = synth_vocabulary(10, 3, seed=3)
    feats = np.random.default_rng(5).integers(0, 256, (500, 32), dtype=np.uint8)
    mb = np.float32(386.1448) / np.float32(718.856)
    tick = [0, 0, 0, 0]

    def nxt(i):
        tick[i] += 1
        return sizes[tick[i] % len(sizes)]

    def capture():                                    # build + capture + replay on fresh handles
        wh = nxt(0); a, b = ORBextractor(*P), ORBextractor(*P)
        r = [extract_pair(a, b, *imgs[wh]) for _ in range(3)]
        assert eq(r[0], r[1]) and eq(r[1], r[2]), "captured chain differs from the plain one"
        return (wh, r[2])

    def first_frame():                                # orbx_reserve: allocations, fills, table uploads
        wh = nxt(1)
        return (wh, ORBextractor(*P)(imgs[wh][0]))

    def vocab():                                      # orbm_vocab_create's uploads + a descent
        v = ORBVocabulary(*voc)
        return v.descend(feats, 1)

    def first_stereo():                               # orbx_stereo_match's first call on fresh handles + a frame's layout download
        wh = nxt(2); a, b = ORBextractor(*P), ORBextractor(*P)
        kl, dl = a(imgs[wh][0]); b(imgs[wh][1])
        u, d = ComputeStereoMatches(a, b, mb, np.float32(386.1448))
        f = Frame(kl, dl, bounds=(0.0, 0.0, float(wh[0]), float(wh[1])))
        return (wh, u, d, f.layout())

    return [capture, first_frame, vocab, first_stereo]


def run_collision(seconds):
    cj = collision_jobs()
    ref = [{} for _ in cj]
    for i, f in enumerate(cj):                        # single-threaded results per size
        for _ in range(4):
            r = f()
            ref[i][repr(r[0]) if i != 2 else "v"] = r
    errs, cnt = [], [0] * len(cj)
    end = time.time() + seconds

    def work(i):
        try:
            while time.time() < end:
                r = cj[i]()
                if not eq(r, ref[i][repr(r[0]) if i != 2 else "v"]):
                    errs.append((i, "result differs from the single-threaded call")); return
                cnt[i] += 1
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(cj))]
    for t in ts: t.start()
    for t in ts: t.join()
    print("collision phase: captures", cnt[0], "first frames", cnt[1], "vocabulary uploads", cnt[2], "first stereo calls", cnt[3], "errors", errs[:5])
    return errs


collision_errors = run_collision(min(seconds, 4.0))
jobs = [make_jobs(k) for k in range(nthreads)]
want = [[f() for _, f in js] for js in jobs]           # single-threaded reference results
errors, counts = [], [0] * nthreads
stop = time.time() + seconds


def worker(k):
    rng = np.random.default_rng(1000 + k)
    try:
        while time.time() < stop:
            j = int(rng.integers(0, len(jobs[k])))
            got = jobs[k][j][1]()
            if not eq(got, want[k][j]):
                errors.append((k, jobs[k][j][0], "result differs from the single-threaded call"))
                return
            counts[k] += 1
    except Exception as e:  # noqa: BLE001
        errors.append((k, repr(e)))


threads = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
for t in threads: t.start()
for t in threads: t.join()
print("threads", nthreads, "seconds", seconds, "calls per thread", counts, "errors", errors[:5])
sys.exit(1 if errors or collision_errors else 0)
