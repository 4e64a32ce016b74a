"""Is the pipelined step loop bound by host-side submission?  (not a test)  Prints the time the Python loop needs to
enqueue K steps and the time until the GPU has finished them, for 1..4 contexts."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from orb_slam2_e_amd import ORBextractor, ORBmatcher
from orb_slam2_e_amd.synth import synth_frames
W, H, BATCH = 640, 480, 64
dev = torch.device("cuda", 0)
d_frames = torch.from_numpy(synth_frames(BATCH, W, H)).to(dev)
m = ORBmatcher(0.6)
qa = torch.arange(BATCH, dtype=torch.int32, device=dev); qb = ((qa + 1) % BATCH).to(torch.int32)
for P in (1, 2, 3, 4):
    ctxs = []
    for i in range(P):
        ex = ORBextractor(2000, 1.2, 8, 20, 7); ts = torch.cuda.Stream(device=dev); st = ts.cuda_stream
        cap = ex.capacity
        bufs = [torch.empty((BATCH, cap), dtype=torch.int32, device=dev) for _ in range(4)]; nm = torch.zeros(BATCH, dtype=torch.int32, device=dev)
        ex.extract_batch_device(d_frames.data_ptr(), BATCH, H, W, st)
        _, desc_p, cnt_p, _ = ex.result_dev()
        ctxs.append((ex, ts, st, bufs, nm, desc_p, cnt_p, cap))
    torch.cuda.synchronize()
    def step(k):
        ex, ts, st, bufs, nm, desc_p, cnt_p, cap = ctxs[k % P]
        ex.extract_batch_device(d_frames.data_ptr(), BATCH, H, W, st)
        m.match_batch_device(desc_p, cnt_p, cap, qa.data_ptr(), qb.data_ptr(), BATCH, bufs[0].data_ptr(), bufs[1].data_ptr(),
                             bufs[2].data_ptr(), bufs[3].data_ptr(), nm.data_ptr(), stream=st)
    for k in range(20): step(k)
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for k in range(K): step(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"contexts {P}: enqueue {1e3*(t1-t0)/K:.3f} ms/step, done {1e3*(t2-t0)/K:.3f} ms/step  ({K*BATCH/(t2-t0):.0f} frames/s)")
    del ctxs
