"""BASELINE config 2 at its full size: a batch of 64 synthetic 640x480 frames through orbx_extract_batch +
orbm_match_batch_dev on the pairs (i, i+1 mod 64) -- exactly bench.py's step, same frames, same launch geometry
(52,160 FAST waves, XCD frame mapping, the 83-MB workspaces) -- against the CPU oracle: keypoints (float bits),
descriptors, best / second / arg-best, accepted matches (TH_LOW, ratio 0.6) and match counts, bit for bit.
ORBextractor.cc:1051-1113, ORBmatcher.cc:645-676."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bench
from orb_slam2_e_amd import ORBextractor, ORBmatcher
from orb_slam2_e_amd.synth import synth_sequence


@pytest.mark.parametrize("kernel", ["matrix cores", "popcount"])
def test_batch64_extract_and_match_equal_oracle(kernel):
    import torch
    B, W, H = bench.BATCH, bench.W, bench.H
    assert B == 64
    frames = synth_sequence(B, W, H)
    d_frames = torch.from_numpy(frames).cuda()
    ex = ORBextractor(*bench.PARAMS)
    m = ORBmatcher(0.6)
    prev = ORBmatcher.set_allpairs_kernel(ORBmatcher.ALLPAIRS_POPCOUNT if kernel == "popcount" else ORBmatcher.ALLPAIRS_AUTO)
    try:
        cap = ex.capacity
        qa = torch.arange(B, dtype=torch.int32).cuda()
        qb = ((qa + 1) % B).to(torch.int32)
        out = [torch.full((B, cap), -7, dtype=torch.int32).cuda() for _ in range(4)]
        nm = torch.zeros(B, dtype=torch.int32).cuda()
        ts = torch.cuda.Stream()            # extraction and matching queued on ONE stream, as in bench.py (a null stream
        st = ts.cuda_stream                 # handle would mean "the handle's own stream" to the extractor only)
        for _ in range(2):          # twice on the same workspace: nothing may survive from the first pass
            ex.extract_batch_device(d_frames.data_ptr(), B, H, W, st)
            kps_p, desc_p, cnt_p, _ = ex.result_dev()
            m.match_batch_device(desc_p, cnt_p, cap, qa.data_ptr(), qb.data_ptr(), B, out[0].data_ptr(), out[1].data_ptr(),
                                 out[2].data_ptr(), out[3].data_ptr(), nm.data_ptr(), stream=st)
        torch.cuda.synchronize()
        kps, desc, cnt = ex.download_batch()
        best, second, idx, m12 = (o.cpu().numpy() for o in out)
        ok, why = bench.verify_against_oracle(frames, kps, desc, cnt, best, second, idx, m12, nm.cpu().numpy(), qb.cpu().numpy())
        assert ok, why
        assert cnt.min() >= 1990 and nm.cpu().numpy().mean() > 300      # the acceptance path sees hundreds of positives per frame
    finally:
        ORBmatcher.set_allpairs_kernel(prev)
