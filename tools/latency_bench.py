"""Single-frame latency of the drop-in call ORBextractor::operator() (host image in, keypoints and descriptors
out: one H2D, the whole kernel chain for a batch of 1, one D2H), the shape a live tracker uses."""
import time
import numpy as np
from orb_slam2_e_amd import ORBextractor
from orb_slam2_e_amd.synth import synth_frame

ex = ORBextractor(2000, 1.2, 8, 20, 7)
imgs = [synth_frame(k) for k in range(8)]
for im in imgs: ex(im)
t = []
for r in range(200):
    t0 = time.perf_counter(); k, d = ex(imgs[r % 8]); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e3
print("operator() 640x480, 2000 features: median %.3f ms  p90 %.3f ms  min %.3f ms  (%d keypoints)" % (np.median(t), np.percentile(t, 90), t.min(), len(k)))
