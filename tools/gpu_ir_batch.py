#!/usr/bin/env python3
"""Developer diagnostic: the RCE Jacobian's radiative work on AdiabatClimate's doubled radiative grid
(src/adiabat/clima_adiabat_solve.f90:768-822: nz_r + 1 IR-only calls on unchanged opacities; nz_r = 2 nz + 2,
src/adiabat/clima_adiabat.f90:729-773) through radtran_radiate_ir_batch, config 2's tables, host arrays in / out.
Usage: gpu_ir_batch.py [nz ...] (AdiabatClimate nz).  CLIMA_HIP_BATCH_SHARED=0 times the per-column form;
CLIMA_BATCH_PIN=0 leaves the result arrays pageable (radtran_batch_pin_results_set)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from clima_amd import synthetic as S
from clima_amd.atmosphere import copy_atm_to_radiative_grid
from clima_amd.radtran import Radtran
PIN = os.environ.get("CLIMA_BATCH_PIN", "1") != "0"    # the caller's result arrays page-locked (the default here: a caller that keeps them); 0: through the pinned block
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
    res = {}
    for mode in (0, 1):      # the general kernel; the response form (radtran_ir_green_set) where the batch qualifies
        r.ir_green = mode
        out = r.radiate_ir_batch(Ts, T)
        best = 1e9
        for rep in range(5):     # (the caller keeps its result arrays, as the Fortran host does)
            t0 = time.time()
            r.radiate_ir_batch(Ts, T, out=out, pin=PIN)
            best = min(best, time.time() - t0)
        res[mode] = (best, out, r.ir_green_batches)
    gen, best = res[0][0], res[1][0]
    dev = max(float(np.max(np.abs(a - b)) / np.max(np.abs(b))) for a, b in zip(res[1][1], res[0][1]))
    # one call at a time, for scale (IR only, stored opacities)
    r.upload_column(*col.args())
    for _ in range(5): r.radiate_resident(False, False)
    r.synchronize()
    t0 = time.time()
    for _ in range(50): r.radiate_resident(False, False)
    r.synchronize()
    one = (time.time() - t0) / 50
    print("AdiabatClimate nz %3d -> %3d layers, %3d IR-only columns: batch %.2f ms (%.1f us/column)%s; general kernel %.2f ms (%.1f us/column), "
          "largest difference %.1e of the rows' maximum; one resident IR-only call %.1f us"
          % (nz, nzr, ncol, best * 1e3, best * 1e6 / ncol, " [response form]" if res[1][2] > res[0][2] else "", gen * 1e3, gen * 1e6 / ncol, dev, one * 1e6), flush=True)
    del r
