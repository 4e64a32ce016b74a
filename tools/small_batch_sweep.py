"""Where the small-batch forms stop paying: extract_batch + download of B resident-size frames (640 x 480, 2000 features), median wall time per
call, under the four combinations of {pyramid in one launch, 512-thread k_octree} on / off.  usage (GPU box): python3 tools/small_batch_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_e_amd import ORBextractor
from orb_slam2_e_amd.synth import synth_sequence

frames = synth_sequence(32)
for B in (1, 2, 3, 4, 6, 8, 16, 32):
    row = []
    for chain, wide in ((0, 0), (64, 0), (0, 64), (64, 64)):
        os.environ["ORBX_PYR_CHAIN_MAX_BATCH"] = str(chain); os.environ["ORBX_OCT_WIDE_MAX_BATCH"] = str(wide)
        ex = ORBextractor(2000, 1.2, 8, 20, 7)
        for _ in range(10): ex.extract_batch(frames[:B]); ex.download_batch()
        t = []
        for _ in range(40):
            t0 = time.perf_counter(); ex.extract_batch(frames[:B]); ex.download_batch(); t.append(time.perf_counter() - t0)
        row.append(np.median(t) * 1e3)
        del ex
    print("B %2d: per-level + 256 threads %.3f ms | chain %.3f | wide octree %.3f | both %.3f" % (B, *row), flush=True)
