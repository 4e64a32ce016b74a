"""Does capturing one step (13 launches) in a HIP graph per context change the pipelined rate?  (not a test)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from orb_slam2_e_amd import ORBextractor, ORBmatcher
from orb_slam2_e_amd.synth import synth_frames
W, H, BATCH, P = 640, 480, 64, 3
dev = torch.device("cuda", 0)
d_frames = torch.from_numpy(synth_frames(BATCH, W, H)).to(dev)
m = ORBmatcher(0.6)
qa = torch.arange(BATCH, dtype=torch.int32, device=dev); qb = ((qa + 1) % BATCH).to(torch.int32)
ctxs = []
for i in range(P):
    ex = ORBextractor(2000, 1.2, 8, 20, 7); ts = torch.cuda.Stream(device=dev); st = ts.cuda_stream
    cap = ex.capacity
    bufs = [torch.empty((BATCH, cap), dtype=torch.int32, device=dev) for _ in range(4)]; nm = torch.zeros(BATCH, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_frames.data_ptr(), BATCH, H, W, st)
    _, desc_p, cnt_p, _ = ex.result_dev()
    ctxs.append(dict(ex=ex, ts=ts, st=st, bufs=bufs, nm=nm, desc_p=desc_p, cnt_p=cnt_p, cap=cap))
torch.cuda.synchronize()
def body(c, st):
    c["ex"].extract_batch_device(d_frames.data_ptr(), BATCH, H, W, st)
    m.match_batch_device(c["desc_p"], c["cnt_p"], c["cap"], qa.data_ptr(), qb.data_ptr(), BATCH, c["bufs"][0].data_ptr(), c["bufs"][1].data_ptr(),
                         c["bufs"][2].data_ptr(), c["bufs"][3].data_ptr(), c["nm"].data_ptr(), stream=st)
def run(step, K=300):
    for k in range(30): step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K): step(k)
    torch.cuda.synchronize()
    return K * BATCH / (time.perf_counter() - t0)
print("plain launches: %.0f frames/s" % run(lambda k: body(ctxs[k % P], ctxs[k % P]["st"])))
for c in ctxs:
    c["g"] = torch.cuda.CUDAGraph()
    with torch.cuda.graph(c["g"], stream=c["ts"]):
        body(c, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
def gstep(k):
    c = ctxs[k % P]
    with torch.cuda.stream(c["ts"]):
        c["g"].replay()
print("graph replay  : %.0f frames/s" % run(gstep))
print("plain launches: %.0f frames/s" % run(lambda k: body(ctxs[k % P], ctxs[k % P]["st"])))
nm_plain = ctxs[0]["nm"].clone(); torch.cuda.synchronize()
