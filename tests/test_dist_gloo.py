"""World-size-2 gloo tests (CPU tensors) of the N > 1 host logic that bench.py runs (orb_slam2_e_amd/shard.py):
frames sharded contiguously per rank, fixed-size records in the bench's own layout, steps rotating over contexts,
the bucketed gather (`--gather-every`, partial buckets flushed before the barrier), MAX-over-ranks timing and the FEM
displacement gather.  The compute leg is a stand-in here: the CPU oracle (checker role only) or a byte pattern."""
import os
import pickle
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FR = 2          # frames per rank
NFEAT, NLEV = 500, 6
CAP = NFEAT + 3 * NLEV
W, H = 320, 240


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _oracle_record(lay, frames):
    """One step's record of `frames` (this rank's shard) in the bench layout, computed by the oracle."""
    import oracle
    o = oracle.OrbOracle(NFEAT, 1.2, NLEV, 20, 7)
    rec = np.zeros(lay.rec_bytes, np.uint8)
    R = lay.unpack(rec)
    descs = []
    for i, f in enumerate(frames):
        kps, desc = o.extract(f)
        n = len(kps)
        R["kps"][i, :n] = kps; R["desc"][i, :n] = desc; R["counts"][i] = n
        descs.append(desc)
    for i in range(len(frames)):
        b, s, ix = oracle.match_bruteforce(descs[i], descs[(i + 1) % len(frames)])
        m12, nm = oracle.match_filter(b, s, ix, 45, 0.6)
        R["match12"][i, :len(m12)] = m12; R["nmatch"][i] = nm
    return rec


def _worker_records(rank, world, port, out):
    _init(rank, world, port)
    from orb_slam2_e_amd.shard import RecordLayout, ShardedPipeline, max_over_ranks
    from orb_slam2_e_amd.synth import synth_sequence
    lay = RecordLayout(FR, CAP)
    frames = synth_sequence(FR, W, H, start=rank * FR)        # rank r owns frames [FR r, FR r + FR)
    got = {}
    state = {}

    def compute(c, k):
        state["rec"] = torch.from_numpy(_oracle_record(lay, frames))

    def pack(c, dst):
        dst.copy_(state["rec"])

    pipe = ShardedPipeline(rank, world, lay.rec_bytes, 1, 1, compute, pack,
                           on_receive=lambda k, r, rec: got.__setitem__((k, r), rec.numpy().copy()))
    pipe.step(0)
    pipe.flush()
    t = max_over_ranks(0.1 * (rank + 1), world)
    assert abs(t - 0.1 * world) < 1e-12
    dist.barrier()
    if rank == 0:
        with open(out, "wb") as f:
            pickle.dump(got, f)
    dist.destroy_process_group()


def test_two_rank_sharding_and_record_layout(tmp_path):
    """Records in bench.py's layout, gathered on rank 0 in global frame order, equal the oracle's results."""
    out = str(tmp_path / "gathered.pkl")
    mp.spawn(_worker_records, args=(2, _free_port(), out), nprocs=2, join=True)
    got = pickle.load(open(out, "rb"))
    sys.path.insert(0, ROOT)
    import oracle
    from orb_slam2_e_amd.shard import RecordLayout
    from orb_slam2_e_amd.synth import synth_sequence
    lay = RecordLayout(FR, CAP)
    assert sorted(got) == [(0, 0), (0, 1)]
    o = oracle.OrbOracle(NFEAT, 1.2, NLEV, 20, 7)
    frames = synth_sequence(2 * FR, W, H)
    for r in range(2):
        R = lay.unpack(got[(0, r)])
        descs = []
        for i in range(FR):
            kps, desc = o.extract(frames[r * FR + i])
            n = int(R["counts"][i])
            assert n == len(kps) and n > 100
            assert np.array_equal(R["kps"][i, :n].view(np.uint8), kps.view(np.uint8)) and np.array_equal(R["desc"][i, :n], desc)
            descs.append(desc)
        for i in range(FR):
            b, s, ix = oracle.match_bruteforce(descs[i], descs[(i + 1) % FR])
            m12, nm = oracle.match_filter(b, s, ix, 45, 0.6)
            assert int(R["nmatch"][i]) == nm and nm > 20 and np.array_equal(R["match12"][i, :len(m12)], m12)


def _pattern(rank, k, n):
    return ((np.arange(n, dtype=np.int64) * 31 + 7919 * k + 104729 * rank) % 251).astype(np.uint8)


def _worker_buckets(rank, world, port, out, nctx, ge, nsteps):
    _init(rank, world, port)
    from orb_slam2_e_amd.shard import ShardedPipeline
    rec_bytes = 1000
    got = []
    state = {}
    used = []

    def compute(c, k):
        used.append(c.index)
        state[c.index] = torch.from_numpy(_pattern(rank, k, rec_bytes))

    def pack(c, dst):
        dst.copy_(state[c.index])

    pipe = ShardedPipeline(rank, world, rec_bytes, nctx, ge, compute, pack, make_context=lambda c: {"i": c.index},
                           on_receive=lambda k, r, rec: got.append((k, r, rec.numpy().copy())))
    for k in range(nsteps):
        pipe.step(k)
    assert used == [k % nctx for k in range(nsteps)]
    before_flush = len(got)
    pipe.flush()
    dist.barrier()
    if rank == 0:
        with open(out, "wb") as f:
            pickle.dump((got, pipe.gathers, before_flush), f)
    else:
        assert not got
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nctx,ge,nsteps,full,partial", [(2, 1, 3, 4, 1, 1), (2, 2, 3, 7, 2, 1), (2, 3, 1, 5, 5, 0), (2, 2, 4, 3, 0, 2),
                                                                (8, 3, 4, 14, 3, 2)])
def test_bucketed_gather(tmp_path, world, nctx, ge, nsteps, full, partial):
    """`--gather-every ge`: a context's records travel in buckets of ge steps, partial buckets leave on flush();
    every (step, rank) record arrives exactly once with the right bytes.  The last case is BASELINE config 4's shape:
    8 ranks, the bench's three contexts and four steps per gather."""
    out = str(tmp_path / "b.pkl")
    mp.spawn(_worker_buckets, args=(world, _free_port(), out, nctx, ge, nsteps), nprocs=world, join=True)
    got, gathers, before_flush = pickle.load(open(out, "rb"))
    assert gathers == full + partial
    assert before_flush == world * ge * full      # what arrived before the flush came in full buckets
    assert sorted((k, r) for k, r, _ in got) == [(k, r) for k in range(nsteps) for r in range(world)]
    for k, r, rec in got:
        assert np.array_equal(rec, _pattern(r, k, 1000)), (k, r)


def _worker_fem(rank, world, port, out):
    _init(rank, world, port)
    from orb_slam2_e_amd.shard import gather_displacements
    x = np.arange(3 * 11, dtype=np.float64).reshape(3, 11) + 1000.0 * rank
    allx = gather_displacements(x, rank, world)
    assert (allx is None) == (rank != 0)
    if rank == 0:
        np.save(out, allx)
    dist.barrier()
    dist.destroy_process_group()


def test_fem_displacement_gather(tmp_path):
    out = str(tmp_path / "x.npy")
    mp.spawn(_worker_fem, args=(2, _free_port(), out), nprocs=2, join=True)
    allx = np.load(out)
    ref = np.concatenate([np.arange(33, dtype=np.float64).reshape(3, 11) + 1000.0 * r for r in range(2)])
    assert np.array_equal(allx, ref)
    from orb_slam2_e_amd.shard import gather_displacements
    assert np.array_equal(gather_displacements(ref, 0, 1), ref)     # N = 1: no collective


def _worker_sweep(rank, world, port, out):
    _init(rank, world, port)
    from orb_slam2_e_amd.shard import ShardedPipeline, gather_sweep
    rec_bytes = 512
    got = []
    state = {}

    def compute(c, k):
        state[c.index] = torch.from_numpy(_pattern(rank, k, rec_bytes))

    def pack(c, dst):
        dst.copy_(state[c.index])

    pipe = ShardedPipeline(rank, world, rec_bytes, 3, 4, compute, pack, on_receive=lambda k, r, rec: got.append((pipe.GE, k, r, rec.numpy().copy())))

    def sync():
        pipe.flush()
        dist.barrier()

    for k in range(5):                     # a partial bucket is pending when the sweep starts
        pipe.step(k)
    ticks = iter(range(1000))
    sweep = gather_sweep(pipe, sync, world, "cpu", values=(1, 4, 16), warm=2, steps=7, clock=lambda: float(next(ticks)))
    assert pipe.GE == 4 and all(c.nfill == 0 for c in pipe.ctxs)
    assert all(c.send.numel() == 4 * rec_bytes for c in pipe.ctxs)
    assert sorted(sweep) == ["1", "16", "4"] and all(abs(v - 1e3 / 7) < 1e-9 for v in sweep.values())    # one clock tick per region
    pipe.step(99); pipe.flush()
    dist.barrier()
    if rank == 0:
        with open(out, "wb") as f:
            pickle.dump(got, f)
    dist.destroy_process_group()


def test_gather_sweep_at_world_8(tmp_path):
    """BASELINE config 4's shape: after the headline region the bench re-runs short regions at --gather-every 1 / 4 / 16 and puts
    the cadence back; every record of every region arrives once, with the right bytes, under the bucket size in force."""
    out = str(tmp_path / "s.pkl")
    mp.spawn(_worker_sweep, args=(8, _free_port(), out), nprocs=8, join=True)
    got = pickle.load(open(out, "rb"))
    by_ge = {}
    for ge, k, r, rec in got:
        assert np.array_equal(rec, _pattern(r, k, 512)), (ge, k, r)
        by_ge.setdefault(ge, []).append((k, r))
    # headline setting: steps 0..4 (their last partial bucket leaves in the first set_gather_every) and step 99
    assert sorted(by_ge[4]) == sorted([(k, r) for k in list(range(5)) + list(range(2)) + list(range(7)) + [99] for r in range(8)])
    for ge in (1, 16):
        assert sorted(by_ge[ge]) == sorted([(k, r) for k in list(range(2)) + list(range(7)) for r in range(8)])
