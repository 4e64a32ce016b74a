"""Isolated timing of the all-pairs matcher on resident descriptor sets (not a test): python tests/perf_match.py [npairs] [n]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from orb_slam2_e_amd.matcher import ORBmatcher
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cap = n + 64
rng = np.random.default_rng(0)
nsets = npairs + 1
desc = torch.from_numpy(rng.integers(0, 256, (nsets, cap, 32), dtype=np.uint8)).cuda()
cnt = torch.from_numpy(rng.integers(n - 40, n + 40, nsets).astype(np.int32)).cuda()
qa = torch.arange(1, nsets, dtype=torch.int32).cuda(); qb = torch.arange(0, nsets - 1, dtype=torch.int32).cuda()
out = [torch.zeros(npairs * cap, dtype=torch.int32).cuda() for _ in range(4)]; nm = torch.zeros(npairs, dtype=torch.int32).cuda()
m = ORBmatcher(0.6)
st = torch.cuda.current_stream().cuda_stream
def call():
    m.match_batch_device(desc.data_ptr(), cnt.data_ptr(), cap, qa.data_ptr(), qb.data_ptr(), npairs, out[0].data_ptr(), out[1].data_ptr(),
                         out[2].data_ptr(), out[3].data_ptr(), nm.data_ptr(), stream=st)
for _ in range(20): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 200
e0.record()
for _ in range(R): call()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / R
pairs = float((cnt[1:].clamp(max=cap).double() * cnt[:-1].clamp(max=cap).double()).sum())
print(f"match_batch {npairs} pairs of ~{n}: {ms*1e3:.1f} us per call (memset + kernel), {pairs/ms/1e6:.1f} G pairs/s, {pairs*512/ms/1e12:.2f} PFLOP/s fp4-equivalent")
