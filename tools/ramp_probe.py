"""How the 64-frame extract + match step's time depends on what the card did just before: after `idle` seconds of nothing, R
back-to-back timed regions of K pipelined steps each (three contexts, as bench.py), every region bracketed by a synchronisation
like bench.py's.  Prints ms per step of region 0, 1, 2, ... -- region 0 is what a `--warmup 5 --steps 20` run measures, the
plateau is what `ms_per_step_conditioned` measures.  usage (GPU box): python3 tools/ramp_probe.py [K=20] [R=30] [idle=1.0] [W=5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orb_slam2_e_amd import ORBextractor, ORBmatcher
from orb_slam2_e_amd.synth import synth_sequence

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
W = int(sys.argv[4]) if len(sys.argv) > 4 else 5
B, H, Wd = 64, 480, 640
P = (2000, 1.2, 8, 20, 7)
dev = torch.device("cuda", 0)
frames = torch.from_numpy(synth_sequence(B, Wd, H)).to(dev)
m = ORBmatcher(0.6)
qa = torch.arange(B, dtype=torch.int32, device=dev); qb = ((qa + 1) % B).to(torch.int32)
ctx = []
for c in range(3):
    ex = ORBextractor(*P); ts = torch.cuda.Stream(device=dev); cap = ex.capacity
    bufs = [torch.empty((B, cap), dtype=torch.int32, device=dev) for _ in range(4)] + [torch.zeros(B, dtype=torch.int32, device=dev)]
    ex.extract_batch_device(frames.data_ptr(), B, H, Wd, ts.cuda_stream)
    ctx.append((ex, ts, bufs, ex.result_dev()))
torch.cuda.synchronize()


def step(k):
    ex, ts, bufs, (kps_p, desc_p, cnt_p, _) = ctx[k % 3]
    ex.extract_batch_device(frames.data_ptr(), B, H, Wd, ts.cuda_stream)
    m.match_batch_device(desc_p, cnt_p, ex.capacity, qa.data_ptr(), qb.data_ptr(), B, *[b.data_ptr() for b in bufs], stream=ts.cuda_stream)


for k in range(30): step(k)
torch.cuda.synchronize()
for trial in range(3):
    time.sleep(idle)
    for k in range(W): step(k)
    torch.cuda.synchronize()
    out = []
    for r in range(R):
        t0 = time.perf_counter()
        for k in range(K): step(k)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / K * 1e3)
    print(f"idle {idle}s, W {W}, K {K}: ms/step by region:", " ".join(f"{x:.4f}" for x in out), flush=True)
