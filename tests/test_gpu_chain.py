"""Chained device calls (orbx_extract_batch -> orbm_match_batch_dev on the extractor's resident results) without a
common stream, and the matcher's guard against counts outside [0, cap].  include/orbslam_hip.h: both calls may be
handed NULL; the extractor then works on its handle's stream and the matcher on the default stream, and the library
orders them by events (orbx_detail::order_after_producer)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from orb_slam2_e_amd import ORBextractor, ORBmatcher
from orb_slam2_e_amd.synth import synth_sequence

PRM = (1000, 1.2, 8, 20, 7)


def _chain(frames_sets, streams, rounds):
    """rounds x (extract set k % 2 -> match pairs (i, i+1)); returns the results of the LAST round."""
    import torch
    B, H, W = frames_sets[0].shape
    d_sets = [torch.from_numpy(f).cuda() for f in frames_sets]
    ex = ORBextractor(*PRM)
    m = ORBmatcher(0.6)
    cap = ex.capacity
    qa = torch.arange(B, dtype=torch.int32).cuda()
    qb = ((qa + 1) % B).to(torch.int32)
    out = [torch.full((B, cap), -7, dtype=torch.int32).cuda() for _ in range(4)]
    nm = torch.zeros(B, dtype=torch.int32).cuda()
    torch.cuda.synchronize()
    for k in range(rounds):
        ex.extract_batch_device(d_sets[k % 2].data_ptr(), B, H, W, streams[0])
        _, desc_p, cnt_p, _ = ex.result_dev()
        m.match_batch_device(desc_p, cnt_p, cap, qa.data_ptr(), qb.data_ptr(), B, out[0].data_ptr(), out[1].data_ptr(),
                             out[2].data_ptr(), out[3].data_ptr(), nm.data_ptr(), stream=streams[1])
    torch.cuda.synchronize()
    kps, desc, cnt = ex.download_batch()
    return kps, desc, cnt, [o.cpu().numpy() for o in out], nm.cpu().numpy()


def test_null_stream_chain_is_ordered():
    """Five rounds alternating between two frame sets with NULL handed to both calls (two different streams inside the
    library), and with two explicit streams that differ: the last round's matches must be those of the last round's
    descriptors -- equal to the one-stream run and to the oracle on one pair."""
    import torch
    sets = [synth_sequence(16, 640, 480, start=0), synth_sequence(16, 640, 480, start=400)]
    s = torch.cuda.Stream()
    ref = _chain(sets, (s.cuda_stream, s.cuda_stream), 5)
    s2 = torch.cuda.Stream()
    for streams in ((None, None), (s.cuda_stream, s2.cuda_stream), (None, s2.cuda_stream)):
        got = _chain(sets, streams, 5)
        assert np.array_equal(got[2], ref[2]), streams
        for f in range(16):     # (rows past a frame's count are unspecified: whatever the allocation held)
            n = int(ref[2][f])
            assert np.array_equal(got[1][f, :n], ref[1][f, :n]) and got[0][f, :n].tobytes() == ref[0][f, :n].tobytes(), (streams, f)
        for a, b in zip(got[3], ref[3]):
            for f in range(16):
                n = int(ref[2][f])
                assert np.array_equal(a[f, :n], b[f, :n]), (streams, f)
        assert np.array_equal(got[4], ref[4])
    n0, n1 = int(ref[2][0]), int(ref[2][1])
    rb, rs, ri = oracle.match_bruteforce(ref[1][0, :n0], ref[1][1, :n1])
    assert np.array_equal(ref[3][0][0, :n0], rb) and np.array_equal(ref[3][2][0, :n0], ri)


@pytest.mark.parametrize("kernel", ["matrix cores", "popcount"])
def test_counts_outside_the_capacity_are_clamped(kernel):
    """counts of -5 and cap + 1000 (what a matcher sees that reads an extractor's counts too early): no fault, and the
    results are those of counts clamped to [0, cap].  Both all-pairs kernels."""
    import torch
    cap, nsets = 200, 3
    rng = np.random.default_rng(5)
    desc = rng.integers(0, 256, (nsets, cap, 32), dtype=np.uint8)
    pairs_a = np.array([0, 1, 2, 1, 2], np.int32)
    pairs_b = np.array([1, 2, 0, 1, 2], np.int32)
    m = ORBmatcher(0.6)
    prev = ORBmatcher.set_allpairs_kernel(ORBmatcher.ALLPAIRS_POPCOUNT if kernel == "popcount" else ORBmatcher.ALLPAIRS_AUTO)
    try:
        res = []
        for counts in ([-5, cap + 1000, 150], [0, cap, 150]):
            d = torch.from_numpy(desc).cuda()
            c = torch.tensor(counts, dtype=torch.int32).cuda()
            qa, qb = torch.from_numpy(pairs_a).cuda(), torch.from_numpy(pairs_b).cuda()
            out = [torch.full((len(pairs_a), cap), -7, dtype=torch.int32).cuda() for _ in range(4)]
            nm = torch.full((len(pairs_a),), -1, dtype=torch.int32).cuda()
            m.match_batch_device(d.data_ptr(), c.data_ptr(), cap, qa.data_ptr(), qb.data_ptr(), len(pairs_a), out[0].data_ptr(),
                                 out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), nm.data_ptr(), th=100)
            torch.cuda.synchronize()
            res.append(([o.cpu().numpy() for o in out], nm.cpu().numpy()))
        clamped = [0, cap, 150]
        for p, a in enumerate(pairs_a):
            n = clamped[a]
            for x, y in zip(res[0][0], res[1][0]):
                assert np.array_equal(x[p, :n], y[p, :n]), p
        assert np.array_equal(res[0][1], res[1][1])
        # and the clamped run is the oracle's
        rb, rs, ri = oracle.match_bruteforce(desc[1], desc[2][:150])
        assert np.array_equal(res[1][0][0][1], rb) and np.array_equal(res[1][0][1][1], rs) and np.array_equal(res[1][0][2][1], ri)
        assert res[1][1][0] == 0            # an empty query set accepts nothing
    finally:
        ORBmatcher.set_allpairs_kernel(prev)
