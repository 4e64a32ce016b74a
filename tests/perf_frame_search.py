"""Wall time of the per-frame matcher calls with and without a resident frame (host arrays in / out, so staging and the
result download are inside), 2000 entries x 2000 keypoints."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from orb_slam2_e_amd import Frame, ORBextractor, ORBmatcher, Points, View
from orb_slam2_e_amd.synth import synth_frame, synth_projection_case, synth_tracking_scene


def t(f, reps=200):
    for _ in range(20): f()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps): f()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


q, qd, qa, takes, kps, desc, bounds, occ, ur = synth_projection_case(0, n=2000, nq=2000, hot=2000)
m = ORBmatcher(0.6, True)
fr = Frame(kps, desc, bounds, ur)
print("search_projection  host arrays %.4f ms   resident frame %.4f ms" % (
    t(lambda: m.search_projection(q, qd, qa, takes, kps, desc, bounds, occ, ur, 95)),
    t(lambda: m.frame_search_projection(fr, q, qd, qa, takes, occ, 95))))
print("search_window      host arrays %.4f ms   resident frame %.4f ms" % (
    t(lambda: m.search_window(q, qd, kps, desc, bounds, occ, ur)), t(lambda: m.frame_search_window(fr, q, qd, occ))))
print("frame create (host arrays, 2000 kps) %.4f ms" % t(lambda: Frame(kps, desc, bounds, ur).close(), 100))
s = synth_tracking_scene(11)
lm = s["last_mp"]
cur = Frame(s["kps"], s["desc"], s["bounds"])
view = View(*s["cam"], s["mb"], s["mbf"], s["log_scale_factor"], s["scale_factors"])
last = Points(s["last_valid"], s["pos"][lm], s["mp_desc"][lm], takes=s["last_takes"], octave=s["last_octave"], angle=s["last_angle"])
print("SearchByProjection(Cur, Last) whole, resident frame: %.4f ms" % t(lambda: m.SearchByProjectionLast(cur, view, s["Tcw"], s["Tlw"], last, s["occupied"], 7.0, True)))
npnt = len(s["pos"])
pts = Points(np.ones(npnt, np.uint8), s["pos"], s["mp_desc"], normal=s["normal"], min_distance=s["mind"], max_distance=s["maxd"], takes=np.ones(npnt, np.uint8))
print("SearchLocalPoints (isInFrustum + SearchByProjection), %d points: %.4f ms" % (npnt, t(lambda: m.SearchByProjectionPoints(cur, view, s["Tcw"], pts, s["occupied"], 1.0))))
ex = ORBextractor(2000, 1.2, 8, 20, 7)
img = synth_frame(3)
ex(img)
print("extract 640x480 (host image -> host kps) %.4f ms; frame from extractor %.4f ms" % (t(lambda: ex(img), 50), t(lambda: Frame.from_extractor(ex, 0, bounds).close(), 100)))
