#!/usr/bin/env python3
"""Developer diagnostic: resident radiate() time against the layer count (ModernEarth tables, 1000 bins, 8 zenith
angles), e.g. to compare launch forms: CLIMA_HIP_NO_HALF=1 python tools/gpu_nz_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
tb = S.modern_earth_tables()
for nz in [int(a) for a in sys.argv[1:]] or [70, 96, 100, 128, 150, 160, 192, 200, 224, 250]:
    col = S.modern_earth_column(nz)
    r = Radtran(tb, nz, 8, 0.15)
    r.upload_column(*col.args())
    for _ in range(100): r.radiate_resident()
    r.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(300): r.radiate_resident()
        r.synchronize()
        best = min(best, (time.perf_counter() - t0) / 300)
    print("nz %4d: %.1f us/call" % (nz, best * 1e6), flush=True)
