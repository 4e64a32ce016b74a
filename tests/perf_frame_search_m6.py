"""SearchByProjection(Cur, Last) on a resident frame, 500 calls (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from orb_slam2_e_amd import Frame, ORBmatcher, Points, View
from orb_slam2_e_amd.synth import synth_tracking_scene
s = synth_tracking_scene(11)
lm = s["last_mp"]
m = ORBmatcher(0.6, True)
cur = Frame(s["kps"], s["desc"], s["bounds"])
view = View(*s["cam"], s["mb"], s["mbf"], s["log_scale_factor"], s["scale_factors"])
last = Points(s["last_valid"], s["pos"][lm], s["mp_desc"][lm], takes=s["last_takes"], octave=s["last_octave"], angle=s["last_angle"])
for _ in range(50): m.SearchByProjectionLast(cur, view, s["Tcw"], s["Tlw"], last, s["occupied"], 7.0, True)
t0 = time.perf_counter()
for _ in range(500): r = m.SearchByProjectionLast(cur, view, s["Tcw"], s["Tlw"], last, s["occupied"], 7.0, True)
print("M6 whole: %.4f ms / call, nmatches %d" % ((time.perf_counter() - t0) / 500 * 1e3, r[2]))
