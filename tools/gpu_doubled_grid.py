#!/usr/bin/env python3
"""Developer diagnostic: per-call device time on AdiabatClimate's doubled radiative grid
(copy_atm_to_radiative_grid, src/adiabat/clima_adiabat.f90:729-773: nz_r = 2*nz + 2, every pair of
layers identical -> pair_reuse), config 2's tables.  Usage: gpu_doubled_grid.py [nz ...] (AdiabatClimate nz)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from clima_amd import synthetic as S
from clima_amd.atmosphere import copy_atm_to_radiative_grid
from clima_amd.radtran import Radtran
tb = S.modern_earth_tables()
for nz in [int(a) for a in sys.argv[1:]] or [50, 100, 200]:
    col = S.Column(copy_atm_to_radiative_grid(S.modern_earth_column(nz)))
    nzr = len(col["T"])
    r = Radtran(tb, nzr, 8, 0.15)
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
    print("AdiabatClimate nz %3d -> radiative grid %3d layers: %.1f us/call | " % (nz, nzr, best * 1e6) +
          ", ".join("%s %.1f" % (n, 1e3 * ms / max(c, 1)) for n, (ms, c) in zip(["prep", "opacity|fused", "twostream", "integrate"], ks) if c), flush=True)
    del r
