"""Development aid: shader-clock time per phase of k_fast_cells / k_describe, summed over waves (needs a library built with
-DORBX_PHASE_TIMING, e.g. orb_slam2_e_amd/lib_ph.so copied over liborbslam_hip.so on the GPU box; never the product build)."""
import ctypes as C, sys
import numpy as np
from orb_slam2_e_amd.extractor import ORBextractor
from orb_slam2_e_amd.synth import synth_sequence

ex = ORBextractor(2000, 1.2, 8, 20, 7)
L = ex._L
frames = synth_sequence(64)
for _ in range(3):
    ex.extract_batch(frames)
ex.download_batch()
acc = np.zeros((2, 65536, 8), np.uint64)
L.orbx_debug_phases.argtypes = [C.c_void_p, C.c_int]
assert L.orbx_debug_phases(acc.ctypes.data, 1) == 0
ex.extract_batch(frames)
ex.download_batch()
assert L.orbx_debug_phases(acc.ctypes.data, 0) == 0
for k, (title, names) in enumerate([("k_fast_cells, per wave", ["tile load", "stage A", "stage B", "zero + score", "nms + emit", "epilogue", "(timer)"]),
                                    ("k_describe, per workgroup (wave 0)", ["lookup", "moments", "atan/sincos", "brief kp0", "brief kp1", "brief kp2", "brief kp3", "tail"])]):
    r = acc[k].astype(np.float64)
    used = r.sum(axis=1) > 0
    r = r[used]
    tot = r.sum(axis=1)
    print(title, "records", len(r), "mean total clk", round(tot.mean()), "median", round(np.median(tot)), "p90", round(np.percentile(tot, 90)))
    for i, n in enumerate(names):
        print("   %-14s mean %8.0f clk  %5.1f %%   median %8.0f" % (n, r[:, i].mean(), 100 * r[:, i].sum() / tot.sum(), np.median(r[:, i])))
