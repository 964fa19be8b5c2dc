#!/usr/bin/env python3
"""Developer diagnostic: kernel time vs number of bins (via bin sharding) on one GPU."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
tb = S.modern_earth_tables()
col = S.modern_earth_column(200)
r = Radtran(tb, 200, 8, 0.15)
names = ["prep", "opacity", "twostream", "integrate"]
for world in (1, 2, 3, 4, 6, 8, 16, 32):
    r.set_bin_shard(0, world)
    r.upload_column(*col.args())
    r.profile(True)
    for _ in range(3): r.radiate_resident()
    r.synchronize(); r.profile_reset()
    for _ in range(20): r.radiate_resident()
    r.synchronize()
    ks = [r.kernel_time(i) for i in range(4)]
    sh = r.bin_shard()
    print("world %2d bins %4d (ir %3d sol %3d) op-waves %5d | " % (world, sh[1], sh[3], sh[5], sh[1] * 200 // 64) +
          ", ".join("%s %.1f" % (n, 1e3 * ms / max(c, 1)) for n, (ms, c) in zip(names, ks)), flush=True)
