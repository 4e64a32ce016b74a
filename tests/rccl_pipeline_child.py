"""Child process of tests/test_gpu_rccl.py: the bench's sharded pipeline (orb_slam2_e_amd/shard.py) on DEVICE tensors over RCCL
with one rank -- process-group set-up as bench.py does it, device-side send / receive buffers, the gather issued under each
context's own stream, the barrier and the MAX all-reduce of the timed region.  Prints RCCL-PIPELINE-OK."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from orb_slam2_e_amd.launch import free_port
from orb_slam2_e_amd.shard import ShardedPipeline, max_over_ranks

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(free_port()))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
REC = 1 << 16
got = {}


def make_context(c):
    class U:
        pass
    u = U()
    u.tstream = torch.cuda.Stream()
    u.buf = torch.zeros(REC, dtype=torch.uint8, device=dev)
    return u


def compute(c, k):
    c.user.buf.copy_(((torch.arange(REC, device=dev) * 7 + k * 13) % 251).to(torch.uint8))   # on the context's stream


def pack(c, dst):
    dst.copy_(c.user.buf, non_blocking=True)


def on_receive(k, r, rec):
    got[(k, r)] = rec.cpu().clone()


pipe = ShardedPipeline(0, 1, REC, 3, 4, compute, pack, make_context=make_context,
                       stream_ctx=lambda c: torch.cuda.stream(c.user.tstream), on_receive=on_receive,
                       send_device=dev, coll_device=dev, gather_single_rank=True)
for k in range(23):
    pipe.step(k)
pipe.flush()
torch.cuda.synchronize()
dist.barrier()
torch.cuda.synchronize()
assert pipe.gathers >= 23 // 4 and len(got) == 23, (pipe.gathers, len(got))
for k in range(23):
    want = ((torch.arange(REC) * 7 + k * 13) % 251).to(torch.uint8)
    assert torch.equal(got[(k, 0)], want), k
t = torch.tensor([2.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == 2.5 and max_over_ranks(1.25, 1, dev) == 1.25
oks = [None]
dist.all_gather_object(oks, (True, 0.0))
assert oks == [(True, 0.0)]
dist.destroy_process_group()
print("RCCL-PIPELINE-OK")
