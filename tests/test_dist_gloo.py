"""World-size-2 gloo test (CPU) of the N>1 host logic of bench.py: frames sharded
contiguously per rank, fixed-size records gathered on rank 0, MAX-over-ranks timing.
The compute leg is replaced by the CPU oracle here (checker role only)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FR = 2          # frames per rank
CAP = 1000 + 24


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from orb_slam2_e_amd.synth import synth_frame
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    rec = np.zeros((FR, 4 + CAP * 32), np.uint8)            # {count, desc[CAP][32]} fixed-size record
    for i in range(FR):
        f = rank * FR + i                                    # rank r owns frames [FR*r, FR*r+FR)
        _, desc = o.extract(synth_frame(f, 320, 240))
        rec[i, :4] = np.frombuffer(np.int32(len(desc)).tobytes(), np.uint8)
        rec[i, 4:4 + 32 * len(desc)] = desc.ravel()
    send = torch.from_numpy(rec.ravel())
    recv = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        allrec = torch.stack(recv).numpy().reshape(world * FR, -1)
        np.save(out, allrec)
        assert abs(float(t) - 0.1 * world) < 1e-12
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather(tmp_path):
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    allrec = np.load(out)
    import oracle
    from orb_slam2_e_amd.synth import synth_frame
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    for f in range(2 * FR):                                  # gathered order = global frame order
        _, desc = o.extract(synth_frame(f, 320, 240))
        n = int(np.frombuffer(allrec[f, :4].tobytes(), np.int32)[0])
        assert n == len(desc) and np.array_equal(allrec[f, 4:4 + 32 * n].reshape(n, 32), desc)
