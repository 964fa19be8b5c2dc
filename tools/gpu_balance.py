#!/usr/bin/env python3
"""Developer diagnostic: device time of every rank's bin shard for N = 2, 4, 8 (one GPU runs the
shards one after the other): how well the shard cost model balances an N-GPU run.
Usage: gpu_balance.py [nz] [nzen]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nzen = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tb = S.modern_earth_tables()
col = S.modern_earth_column(nz)
r = Radtran(tb, nz, nzen, 0.15)
for world in (1, 2, 4, 8):
    ts = []
    for rank in range(world):
        r.set_bin_shard(rank, world)
        r.upload_column(*col.args())
        for _ in range(5): r.radiate_resident()
        r.synchronize()
        t0 = time.time()
        for _ in range(100): r.radiate_resident()
        r.synchronize()
        sh = r.bin_shard()
        ts.append(((time.time() - t0) / 100 * 1e6, sh[1], sh[3], sh[5]))
    print("nz %d world %d: max %.1f us | " % (nz, world, max(t[0] for t in ts)) +
          "  ".join("%.0f(b%d i%d s%d)" % t for t in ts), flush=True)
r.set_bin_shard(0, 1)
