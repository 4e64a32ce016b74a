"""Kernel-trace target: 64 resident 1242x375 pairs, left + right extract_batch and one orbx_stereo_match, 10 times.
usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/stereo_batch_prof.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orb_slam2_e_amd import ORBextractor, stereo_download_batch, stereo_match_batch
from orb_slam2_e_amd.synth import synth_stereo_pair
P = (2000, 1.2, 8, 20, 7)
B = 64
pairs = [synth_stereo_pair(100 + k) for k in range(4)]
dl = torch.from_numpy(np.stack([pairs[k % 4][0] for k in range(B)])).cuda()
dr = torch.from_numpy(np.stack([pairs[k % 4][1] for k in range(B)])).cuda()
ts = torch.cuda.Stream(); st = ts.cuda_stream
eL, eR = ORBextractor(*P), ORBextractor(*P)
H, W = pairs[0][0].shape
mb = np.float32(386.1448) / np.float32(718.856)
def step():
    eL.extract_batch_device(dl.data_ptr(), B, H, W, st); eR.extract_batch_device(dr.data_ptr(), B, H, W, st)
    stereo_match_batch(eL, eR, mb, np.float32(386.1448), st)
for _ in range(3): step()
ts.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
ts.synchronize()
print("ms per 64 pairs", (time.perf_counter() - t0) / 10 * 1e3)
