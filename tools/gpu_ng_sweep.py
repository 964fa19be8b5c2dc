#!/usr/bin/env python3
"""Developer diagnostic: per-call time against the number of g-points (config 2's shape: nz = 200,
1000 bins, 5 k-species), ng = 8 (lane-per-item kernel in the fused grid), 16 and 32
(k_opacity_coop<16/32>), others (k_opacity_generic)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
for ng in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 12]:
    tb = S.make_tables(ng=ng)
    col = S.modern_earth_column(200)
    r = Radtran(tb, 200, 8, 0.15)
    r.upload_column(*col.args())
    n = 200 if ng == 8 else 20
    for _ in range(3): r.radiate_resident()
    r.synchronize()
    t0 = time.time()
    for _ in range(n): r.radiate_resident()
    r.synchronize()
    dt = (time.time() - t0) / n
    r.profile(True); r.profile_reset()
    for _ in range(5): r.radiate_resident()
    r.synchronize()
    ks = [r.kernel_time(i) for i in range(4)]
    print("ng %2d: %.1f us/call | " % (ng, dt * 1e6) + ", ".join("%s %.1f" % (nm, 1e3 * ms / max(c, 1)) for nm, (ms, c) in zip(["prep", "opacity|fused", "twostream", "integrate"], ks) if c), flush=True)
    del r
