"""The all-pairs matcher launched back to back on one set of 64 extracted frames (no other kernel between the launches):
per-launch durations come from `rocprofv3 --kernel-trace -- python3 tools/match_alone.py [launches] [gap_ms]`; the script also
prints the event-timed mean and a checksum of the outputs (A/B of kernel variants: ORBX_LIB=path selects the library)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from orb_slam2_e_amd import ORBextractor, ORBmatcher
from orb_slam2_e_amd.synth import synth_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device("cuda:0")
frames = synth_sequence(bench.BATCH, bench.W, bench.H)
d_frames = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
ex = ORBextractor(*bench.PARAMS)
st = torch.cuda.current_stream().cuda_stream
ex.extract_batch_device(d_frames.data_ptr(), bench.BATCH, bench.H, bench.W, st)
kps_p, desc_p, cnt_p, _ = ex.result_dev()
cap = ex.capacity
m = ORBmatcher()
qa = torch.arange(bench.BATCH, dtype=torch.int32, device=dev); qb = ((qa + 1) % bench.BATCH).to(torch.int32)
best = torch.empty((bench.BATCH, cap), dtype=torch.int32, device=dev)
second = torch.empty_like(best); idx = torch.empty_like(best); m12 = torch.empty_like(best)
nm = torch.zeros(bench.BATCH, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(n):
    m.match_batch_device(desc_p, cnt_p, cap, qa.data_ptr(), qb.data_ptr(), bench.BATCH, best.data_ptr(), second.data_ptr(),
                         idx.data_ptr(), m12.data_ptr(), nm.data_ptr(), stream=st)
    if gap > 0:
        torch.cuda.synchronize(); time.sleep(gap * 1e-3)
e1.record()
torch.cuda.synchronize()
import zlib
sig = zlib.crc32(best.cpu().numpy().tobytes() + second.cpu().numpy().tobytes() + idx.cpu().numpy().tobytes() + m12.cpu().numpy().tobytes())
clk = best[:, cap - 1].cpu().numpy()
print("done", n, "row cap-1 of best (cycle stamps in the instrumented builds): mean %.0f" % clk.mean(), "ms per launch (events, back to back) %.4f" % (e0.elapsed_time(e1) / n), "crc32 of best/second/idx/match12 %08x" % sig)
