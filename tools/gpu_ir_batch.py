#!/usr/bin/env python3
"""Developer diagnostic: the RCE Jacobian's radiative work on AdiabatClimate's doubled radiative grid
(src/adiabat/clima_adiabat_solve.f90:768-822: nz_r + 1 IR-only calls on unchanged opacities; nz_r = 2 nz + 2,
src/adiabat/clima_adiabat.f90:729-773) through radtran_radiate_ir_batch, config 2's tables, host arrays in / out.
Usage: gpu_ir_batch.py [nz ...] (AdiabatClimate nz).  CLIMA_HIP_BATCH_SHARED=0 times the per-column form."""
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
    r = Radtran(tb, nzr, 4, 0.15)
    r.radiate(*col.args())
    ncol = nzr + 1
    T = np.repeat(np.asarray(col["T"])[:, None], ncol, axis=1)
    Ts = np.full(ncol, float(col["T_surface"]))
    Ts[0] += 1.0
    for c in range(1, ncol):
        T[c - 1, c] += 1.0
    r.radiate_ir_batch(Ts, T)
    best = 1e9
    for rep in range(3):
        t0 = time.time()
        r.radiate_ir_batch(Ts, T)
        best = min(best, time.time() - t0)
    # one call at a time, for scale (IR only, stored opacities)
    r.upload_column(*col.args())
    for _ in range(5): r.radiate_resident(False, False)
    r.synchronize()
    t0 = time.time()
    for _ in range(50): r.radiate_resident(False, False)
    r.synchronize()
    one = (time.time() - t0) / 50
    print("AdiabatClimate nz %3d -> %3d layers, %3d IR-only columns: batch %.2f ms (%.1f us/column); one resident IR-only call %.1f us"
          % (nz, nzr, ncol, best * 1e3, best * 1e6 / ncol, one * 1e6), flush=True)
    del r
