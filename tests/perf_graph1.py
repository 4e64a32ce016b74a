"""Single frame: device-side latency of the 12-kernel extraction sequence, plain launches vs one HIP graph replay (not a test)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from orb_slam2_e_amd import ORBextractor
from orb_slam2_e_amd.synth import synth_frames
W, H = 640, 480
dev = torch.device("cuda", 0)
d_frames = torch.from_numpy(synth_frames(1, W, H)).to(dev)
ex = ORBextractor(2000, 1.2, 8, 20, 7)
ts = torch.cuda.Stream(device=dev); st = ts.cuda_stream
ex.extract_batch_device(d_frames.data_ptr(), 1, H, W, st); torch.cuda.synchronize()
def plain():
    ex.extract_batch_device(d_frames.data_ptr(), 1, H, W, st)
    torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=ts):
    ex.extract_batch_device(d_frames.data_ptr(), 1, H, W, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
def graph():
    with torch.cuda.stream(ts):
        g.replay()
    torch.cuda.synchronize()
for name, f in (("plain", plain), ("graph", graph), ("plain", plain), ("graph", graph)):
    for _ in range(30): f()
    t = []
    for _ in range(200):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    t.sort(); print(f"{name}: median {1e3*t[100]:.3f} ms  p90 {1e3*t[180]:.3f}  min {1e3*t[0]:.3f}")
