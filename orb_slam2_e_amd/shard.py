"""Multi-GPU host logic of the frames/s pipeline: one process per GPU, rank r owns frames
[B r, B r + B) (and meshes likewise), no data-path collective, fixed-size result records gathered on rank 0.

This module is what `bench.py` runs for N > 1 and what `tests/test_dist_gloo.py` drives on CPU tensors over gloo with
a stand-in compute leg: the record layout, the rotation of steps over independent contexts, the bucketed gather
(`gather_every` steps of a context travel in one collective; partial buckets are flushed before every barrier) and the
displacement gather of the FEM leg.  It knows nothing about HIP: the compute leg and the record packing are callbacks.
"""
import contextlib

import numpy as np
import torch
import torch.distributed as dist


class RecordLayout:
    """One step's record of one rank: {kps[B][cap] (28 B cv::KeyPoint), desc[B][cap][32], counts[B] i32,
    match12[B][cap] i32, nmatch[B] i32}, contiguous in that order (B = frames per rank, cap = nfeatures + 3 nlevels)."""

    def __init__(self, batch, cap):
        self.batch, self.cap = batch, cap
        self.n_kps, self.n_desc, self.n_cnt, self.n_m12 = cap * 28 * batch, cap * 32 * batch, 4 * batch, cap * 4 * batch
        self.o_kps = 0
        self.o_desc = self.o_kps + self.n_kps
        self.o_cnt = self.o_desc + self.n_desc
        self.o_m12 = self.o_cnt + self.n_cnt
        self.o_nm = self.o_m12 + self.n_m12
        self.rec_bytes = self.o_nm + self.n_cnt

    def unpack(self, rec):
        """rec: uint8 array of rec_bytes -> dict of typed numpy views."""
        from .extractor import KP_DTYPE
        rec = np.asarray(rec, np.uint8)
        assert rec.size == self.rec_bytes
        B, cap = self.batch, self.cap
        return {"kps": rec[self.o_kps:self.o_desc].view(KP_DTYPE).reshape(B, cap),
                "desc": rec[self.o_desc:self.o_cnt].reshape(B, cap, 32),
                "counts": rec[self.o_cnt:self.o_m12].view(np.int32),
                "match12": rec[self.o_m12:self.o_nm].view(np.int32).reshape(B, cap),
                "nmatch": rec[self.o_nm:self.rec_bytes].view(np.int32)}


class Context:
    """One independent pipeline context: its own workspace / stream on the compute side (`user`), a send buffer holding
    up to `gather_every` records, and on rank 0 one receive buffer per rank."""

    def __init__(self, index):
        self.index = index
        self.user = None
        self.send = None
        self.recv = None
        self.nfill = 0
        self.steps = []      # global step numbers of the records waiting in `send`


class ShardedPipeline:
    """step(k) runs the compute leg of step k on context k % P, packs its record behind it and, every `gather_every`
    steps of that context, sends the bucket to rank 0.  All ranks call step()/flush() with the same k sequence.

    compute(ctx, k)          -- enqueue the step's work (on ctx's stream)
    pack(ctx, dst_u8)        -- enqueue the copy of the step's record into dst_u8 (a rec_bytes slice of ctx.send)
    stream_ctx(ctx)          -- context manager making ctx's stream current (nullcontext on CPU)
    on_receive(k, rank, rec) -- rank 0 only, optional: called per received record (a torch uint8 tensor on cdev);
                                 the buffer is reused by the next gather of that context
    send_device / recv_device: where ctx.send and the collective's payload live (same device for RCCL; for a gloo
    rehearsal the payload is staged through the CPU)."""

    def __init__(self, rank, world, rec_bytes, ncontexts, gather_every, compute, pack, make_context=None,
                 stream_ctx=None, on_receive=None, send_device="cpu", coll_device="cpu", enable_gather=True,
                 gather_single_rank=False):
        self.rank, self.world, self.rec_bytes = rank, world, rec_bytes
        self.GE = max(1, int(gather_every))
        self.compute, self.pack = compute, pack
        self.stream_ctx = stream_ctx or (lambda ctx: contextlib.nullcontext())
        self.on_receive = on_receive
        # gather_single_rank: run the collective with one rank too (tests: the RCCL call path on a one-GPU box)
        self.do_gather = (world > 1 or gather_single_rank) and enable_gather
        self.send_device, self.coll_device = torch.device(send_device), torch.device(coll_device)
        self.ctxs = []
        for i in range(max(1, ncontexts)):
            c = Context(i)
            if make_context is not None:
                c.user = make_context(c)
            if self.do_gather:
                c.send = torch.empty(self.GE * rec_bytes, dtype=torch.uint8, device=self.send_device)
                if rank == 0:
                    c.recv = [torch.empty(self.GE * rec_bytes, dtype=torch.uint8, device=self.coll_device) for _ in range(world)]
            self.ctxs.append(c)
        self.gathers = 0

    def gather_bucket(self, c):
        """The records of the last c.nfill steps of this context -> rank 0 (every rank holds the same number)."""
        n = c.nfill * self.rec_bytes
        if not self.do_gather or n == 0:
            return
        with self.stream_ctx(c):
            recv = [r[:n] for r in c.recv] if self.rank == 0 else None
            payload = c.send[:n]
            if payload.device != self.coll_device:
                payload = payload.to(self.coll_device)      # gloo rehearsal: staged through the host (synchronising copy)
            dist.gather(payload, recv, dst=0)
            if self.rank == 0 and self.on_receive is not None:
                for r in range(self.world):
                    for j, k in enumerate(c.steps):
                        self.on_receive(k, r, c.recv[r][j * self.rec_bytes:(j + 1) * self.rec_bytes])
        self.gathers += 1
        c.nfill = 0
        c.steps = []

    def step(self, k):
        c = self.ctxs[k % len(self.ctxs)]
        with self.stream_ctx(c):
            self.compute(c, k)
            if self.do_gather:
                base = c.nfill * self.rec_bytes
                self.pack(c, c.send[base:base + self.rec_bytes])
                c.nfill += 1
                c.steps.append(k)
        if self.do_gather and c.nfill == self.GE:
            self.gather_bucket(c)

    def flush(self):
        """Partial buckets: every step's records are on rank 0 before the caller's barrier."""
        for c in self.ctxs:
            self.gather_bucket(c)

    def set_gather_every(self, gather_every, sync=None):
        """Changes the bucket size between timed regions (all ranks, same value): flushes, waits (`sync`: the caller's
        barrier + device synchronisation -- the buffers of a collective still in flight must not be released), then re-makes
        the send / receive buffers for the new size."""
        self.flush()
        if sync is not None:
            sync()
        self.GE = max(1, int(gather_every))
        if not self.do_gather:
            return
        for c in self.ctxs:
            c.send = torch.empty(self.GE * self.rec_bytes, dtype=torch.uint8, device=self.send_device)
            if self.rank == 0:
                c.recv = [torch.empty(self.GE * self.rec_bytes, dtype=torch.uint8, device=self.coll_device) for _ in range(self.world)]


def gather_sweep(pipe, sync, world, coll_device, values=(1, 4, 16), warm=6, steps=24, clock=None):
    """Short timed regions at several bucket sizes (after the headline region, which keeps its own setting): ms per step, MAX
    over ranks, per value -- one multi-GPU run then shows which cadence this node wants.  Restores the pipeline's setting."""
    import time
    clock = clock or time.perf_counter
    keep = pipe.GE
    out = {}
    for ge in values:
        pipe.set_gather_every(ge, sync)
        for k in range(warm):
            pipe.step(k)
        sync()
        t0 = clock()
        for k in range(steps):
            pipe.step(k)
        sync()
        out[str(ge)] = max_over_ranks(clock() - t0, world, coll_device) / steps * 1e3
    pipe.set_gather_every(keep, sync)
    return out


def max_over_ranks(seconds, world, device="cpu"):
    """The timed region's duration as the driver wants it: MAX over ranks."""
    if world <= 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_displacements(x_local, rank, world, coll_device="cpu"):
    """FEM leg ("FEM: gather displacements"): x_local [meshes_per_rank, ndof] f64 on every rank ->
    [world * meshes_per_rank, ndof] on rank 0 (mesh m of rank r at row r * meshes_per_rank + m), None elsewhere."""
    x = torch.as_tensor(np.ascontiguousarray(x_local, np.float64)) if not torch.is_tensor(x_local) else x_local
    if world <= 1:
        return x.cpu().numpy()
    x = x.to(coll_device).contiguous()
    recv = [torch.empty_like(x) for _ in range(world)] if rank == 0 else None
    dist.gather(x, recv, dst=0)
    return torch.cat(recv, 0).cpu().numpy() if rank == 0 else None
