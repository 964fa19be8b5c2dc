#!/usr/bin/env python3
"""Developer diagnostic: per-call device time against the number of layers (config 2's tables)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
tb = S.modern_earth_tables()
for nz in [int(a) for a in sys.argv[1:]] or [50, 102, 128, 200, 256]:
    col = S.modern_earth_column(nz)
    r = Radtran(tb, nz, 8, 0.15)
    r.upload_column(*col.args())
    best = 1e9
    for rep in range(3):
        for _ in range(20): r.radiate_resident()
        r.synchronize()
        t0 = time.time()
        for _ in range(200): r.radiate_resident()
        r.synchronize()
        best = min(best, (time.time() - t0) / 200)
    r.profile(True); r.profile_reset()
    for _ in range(30): r.radiate_resident()
    r.synchronize()
    ks = [r.kernel_time(i) for i in range(4)]
    print("nz %4d: %.1f us/call | " % (nz, best * 1e6) + ", ".join("%s %.1f" % (n, 1e3 * ms / max(c, 1)) for n, (ms, c) in zip(["prep", "opacity|fused", "twostream", "integrate"], ks) if c), flush=True)
    del r
