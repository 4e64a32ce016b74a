"""Where the latency of ONE frame through ORBextractor::operator() goes (the shape a live tracker uses: batch of 1): the call's wall time
(median of 200), and the kernel time by kind from the library's event profiler in a separate pass.  usage (GPU box): python3 tools/single_frame_prof.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_e_amd import ORBextractor
from orb_slam2_e_amd._lib import lib
from orb_slam2_e_amd.synth import synth_frame

# usage: single_frame_prof.py [w h nfeatures scale nlevels iniTh minTh]   (this fork's launch files: 640 360 1200 1.1 6 24 7)
A = sys.argv[1:]
W_, H_ = (int(A[0]), int(A[1])) if len(A) >= 2 else (640, 480)
PRM = (int(A[2]), float(A[3]), int(A[4]), int(A[5]), int(A[6])) if len(A) >= 7 else (2000, 1.2, 8, 20, 7)
ex = ORBextractor(*PRM)
imgs = [synth_frame(k, w=W_, h=H_) for k in range(8)]
for im in imgs: ex(im)
t = []
for r in range(200):
    t0 = time.perf_counter(); k, d = ex(imgs[r % 8]); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e3
print("operator() %dx%d, %s: median" % (W_, H_, PRM) + " %.3f ms  p90 %.3f ms  min %.3f ms  (%d keypoints)" % (np.median(t), np.percentile(t, 90), t.min(), len(k)))
L = lib()
L.orbx_profile_enable(ex._h, -1)
for r in range(50): ex(imgs[r % 8])
names = (C.c_char_p * 16)(); ms = (C.c_double * 16)(); n = (C.c_int64 * 16)(); nk = C.c_int()
L.orbx_profile_read(ex._h, 16, names, ms, n, C.byref(nk))
tot = 0.0
for i in range(nk.value):
    if n[i]:
        print("  %-16s %3d launches per frame, %.4f ms per frame" % (names[i].decode(), n[i] // 50, ms[i] / 50)); tot += ms[i] / 50
print("  kernels in sum %.4f ms per frame (events around every launch: the launches do not overlap in this pass)" % tot)
