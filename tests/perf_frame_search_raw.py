"""The library's share of a resident-frame call: the C entry point called in a loop with arguments prepared once."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from orb_slam2_e_amd import Frame, ORBmatcher, Points, View
from orb_slam2_e_amd._lib import lib
from orb_slam2_e_amd.synth import synth_tracking_scene, synth_projection_case
L = lib()
p = lambda a: a.ctypes.data_as(C.c_void_p)
s = synth_tracking_scene(11)
lm = s["last_mp"]
cur = Frame(s["kps"], s["desc"], s["bounds"])
view = View(*s["cam"], s["mb"], s["mbf"], s["log_scale_factor"], s["scale_factors"])
last = Points(s["last_valid"], s["pos"][lm], s["mp_desc"][lm], takes=s["last_takes"], octave=s["last_octave"], angle=s["last_angle"])
Tc = np.ascontiguousarray(s["Tcw"], np.float32).reshape(16); Tl = np.ascontiguousarray(s["Tlw"], np.float32).reshape(16)
occ = np.ascontiguousarray(s["occupied"], np.uint8)
mk = np.zeros(cur.n, np.int32); mq = np.zeros(last.n, np.int32); nm = C.c_int(0)
fn = L.orbm_search_by_projection_last
args = (cur._h, C.byref(view.c), p(Tc), p(Tl), C.byref(last.c), p(occ), C.c_float(7.0), 1, 95, 1, p(mk), p(mq), C.byref(nm), None)
def t(f, reps=500):
    for _ in range(50): f()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps): f()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best
print("orbm_search_by_projection_last raw: %.4f ms (nmatches %d)" % (t(lambda: fn(*args)), nm.value))
q, qd, qa, takes, kps, desc, bounds, occ2, ur = synth_projection_case(0, n=2000, nq=2000, hot=2000)
fr = Frame(kps, desc, bounds, ur)
q = np.ascontiguousarray(q, ORBmatcher.WQ_DTYPE)
mk2 = np.zeros(fr.n, np.int32); mq2 = np.zeros(len(q), np.int32)
fn2 = L.orbm_frame_search_projection
fn2.argtypes = None
a2 = (fr._h, p(q), p(qd), p(qa), p(takes), len(q), p(occ2), 95, C.c_float(0.6), 0, 1, p(mk2), p(mq2), C.byref(nm))
print("orbm_frame_search_projection raw: %.4f ms (nmatches %d)" % (t(lambda: fn2(*a2)), nm.value))
