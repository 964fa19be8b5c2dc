#!/usr/bin/env python3
"""Developer diagnostic: calls shaped like AdiabatClimate's (templates/AdiabatClimate: 50 layers -> 102-layer
doubled radiative grid, 4 zenith angles) at several bin counts: synchronous TOA_fluxes and resident calls,
per-kernel device time.  Usage: gpu_adiabat_like.py [nw ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from clima_amd import synthetic as S
from clima_amd.atmosphere import copy_atm_to_radiative_grid
from clima_amd.radtran import Radtran
for nw in [int(a) for a in sys.argv[1:]] or [100, 200, 400, 1000]:
    tb = S.modern_earth_tables(nw=nw)
    col = S.Column(copy_atm_to_radiative_grid(S.modern_earth_column(50)))
    nzr = len(col["T"])
    r = Radtran(tb, nzr, 4, 0.3)
    a = col.args()
    for _ in range(10): r.TOA_fluxes(*a)
    ts = []
    for _ in range(100):
        t0 = time.perf_counter(); r.TOA_fluxes(*a); ts.append(time.perf_counter() - t0)
    r.bench_toa_fluxes(10, *a)
    tc = r.bench_toa_fluxes(100, *a)      # the same call timed inside the library (no ctypes layer)
    r.upload_column(*a)
    for _ in range(20): r.radiate_resident()
    r.synchronize()
    t0 = time.time()
    for _ in range(200): r.radiate_resident()
    r.synchronize()
    res = (time.time() - t0) / 200
    r.profile(True); r.profile_reset()
    for _ in range(30): r.radiate_resident()
    r.synchronize()
    ks = [r.kernel_time(i) for i in range(4)]
    print("nw %4d x %d layers (%d source layers): sync TOA_fluxes median %.1f us (inside the library: %.1f us), resident %.1f us/call | " % (nw, nzr, nzr // 2, 1e6 * np.median(ts), float(np.median(tc)), res * 1e6) +
          ", ".join("%s %.1f" % (n, 1e3 * ms / max(c, 1)) for n, (ms, c) in zip(["prep", "opacity|fused", "twostream", "integrate"], ks) if c), flush=True)
    del r
