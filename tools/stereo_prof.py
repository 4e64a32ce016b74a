"""Config 5 pair in a loop (for rocprofv3 --kernel-trace): left/right extract + ComputeStereoMatches."""
import time
import numpy as np
from orb_slam2_e_amd import ComputeStereoMatches, ORBextractor
from orb_slam2_e_amd.synth import synth_stereo_pair
P = (2000, 1.2, 8, 20, 7)
mb = np.float32(386.1448) / np.float32(718.856)
left, right = synth_stereo_pair(0)
eL, eR = ORBextractor(*P), ORBextractor(*P)
for _ in range(30):
    eL(left); eR(right); ComputeStereoMatches(eL, eR, mb, np.float32(386.1448))
t0 = time.perf_counter()
for _ in range(50): ComputeStereoMatches(eL, eR, mb, np.float32(386.1448))
print("ComputeStereoMatches %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
