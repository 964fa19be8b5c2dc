"""N>1 path on CPU: world_size-2 `gloo` run of the bin-sharded reduction.

Each rank owns a contiguous, work-balanced range of opacity bins (clima_amd.sharding),
integrates its bins' spectra over frequency, and ONE all-reduce of the 4*(nz+1) partial
level fluxes reproduces the unsharded result.  The per-bin spectra come from the oracle
here (the checker; no GPU in this test) -- what is under test is the partition and the
torch.distributed plumbing bench.py uses.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clima_amd import synthetic as S
    from clima_amd.sharding import bin_shard
    from oracle import oracle as O
    tb = S.modern_earth_tables(nw=40)
    nz, nzen = 20, 2
    col = S.modern_earth_column(nz)
    o = O.OracleRadtran(tb, nz, nzen, 0.3)
    o.radiate(*col.args())
    ir0 = o.ir_start
    op_lo, op_n, ir_lo, ir_n, sol_lo, sol_n = bin_shard(tb.nw, (ir0, ir0 + o.nw_ir - 1), (0, o.nw_sol - 1), nzen,
                                                         rank, world)
    part = torch.zeros(4, nz + 1, dtype=torch.float64)
    for a, (w, fr, lo, n) in enumerate(((o.wrk_ir, None, ir_lo, ir_n), (o.wrk_ir, None, ir_lo, ir_n),
                                        (o.wrk_sol, None, sol_lo, sol_n), (o.wrk_sol, None, sol_lo, sol_n))):
        wavl = tb.ir_wavl if a < 2 else tb.sol_wavl
        freq = 299792458.0 / (wavl * 1e-9)
        dfreq = freq[:-1] - freq[1:]
        arr = w.fup_a if a % 2 == 0 else w.fdn_a
        part[a] = torch.from_numpy(arr[:, lo:lo + n] @ dfreq[lo:lo + n])
    dist.all_reduce(part)                       # the single collective of the sharded path
    full = np.stack([o.wrk_ir.fup_n, o.wrk_ir.fdn_n, o.wrk_sol.fup_n, o.wrk_sol.fdn_n])
    ok = np.allclose(part.numpy(), full, rtol=1e-12, atol=1e-9)
    counts = torch.tensor([op_n], dtype=torch.int64)
    dist.all_reduce(counts)
    q.put((rank, bool(ok), int(counts[0]), op_lo, op_n))
    dist.destroy_process_group()


def test_two_rank_bin_sharded_allreduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _, _ in res)
    assert all(total == 40 for _, _, total, _, _ in res)            # every bin owned exactly once
    assert res[0][3] == 0 and res[0][3] + res[0][4] == res[1][3]     # contiguous ranges


def test_partition_is_balanced_and_complete():
    from clima_amd.sharding import bin_costs, bin_shard
    nw, ir, sol, nzen = 1000, (400, 999), (0, 599), 8
    cost = np.array(bin_costs(nw, ir, sol, nzen))
    for world in (1, 2, 4, 8):
        shards = [bin_shard(nw, ir, sol, nzen, r, world) for r in range(world)]
        assert shards[0][0] == 0 and sum(s[1] for s in shards) == nw
        for a, b in zip(shards, shards[1:]):
            assert a[0] + a[1] == b[0]
        loads = [cost[s[0]:s[0] + s[1]].sum() for s in shards]
        assert max(loads) <= 1.05 * cost.sum() / world + cost.max()
        assert sum(s[3] for s in shards) == 600 and sum(s[5] for s in shards) == 600


def _id_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clima_amd.radtran import Radtran
    # what bench.py does for N > 1: rank 0 draws the id of the LIBRARY's communicator, torch.distributed hands it round
    ids = [Radtran.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    q.put((rank, bytes(ids[0])))
    dist.destroy_process_group()


def test_communicator_id_reaches_every_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_id_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(res[0]) == 128 and res[0] == res[1] and any(res[0])
