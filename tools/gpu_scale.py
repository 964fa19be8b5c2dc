#!/usr/bin/env python3
"""Developer diagnostic: per-call device time vs number of bins (via bin sharding: what one rank
of an N-GPU run executes) on one GPU.  Usage: gpu_scale.py [nz] [nzen]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nzen = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tb = S.modern_earth_tables()
col = S.modern_earth_column(nz)
r = Radtran(tb, nz, nzen, 0.15)
names = ["prep", "opacity|fused", "twostream", "integrate"]
for fused in (True, False):
    r.fused = fused
    for world in (1, 2, 4, 8):
        r.set_bin_shard(0, world)
        r.upload_column(*col.args())
        r.profile(False)
        for _ in range(5): r.radiate_resident()
        r.synchronize()
        t0 = time.time()
        for _ in range(100): r.radiate_resident()
        r.synchronize()
        dt = (time.time() - t0) / 100
        r.profile(True); r.profile_reset()
        for _ in range(20): r.radiate_resident()
        r.synchronize()
        ks = [r.kernel_time(i) for i in range(4)]
        sh = r.bin_shard()
        print("nz %d fused %d world %d bins %4d (ir %3d sol %3d) %.1f us/call | " % (nz, fused, world, sh[1], sh[3], sh[5], dt * 1e6) +
              ", ".join("%s %.1f" % (n, 1e3 * ms / max(c, 1)) for n, (ms, c) in zip(names, ks) if c), flush=True)
r.set_bin_shard(0, 1)
