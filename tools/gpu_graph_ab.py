#!/usr/bin/env python3
"""Developer diagnostic: what a hipGraph makes of a resident call's launches (clima_bench_resident_graph: one call captured,
replayed; timing only) beside the plain launches, per call with a synchronise each and 20 calls back to back.
Config 2 and an AdiabatClimate-shaped call (102-layer doubled grid, 400 bins, 4 zenith angles)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from clima_amd import synthetic as S
from clima_amd.atmosphere import copy_atm_to_radiative_grid
from clima_amd.radtran import Radtran
cases = [("config 2 (200 layers, 1000 bins, 8 zenith angles)", S.modern_earth_tables(), S.modern_earth_column(200), 8),
         ("102-layer doubled grid, 400 bins, 4 zenith angles", S.modern_earth_tables(nw=400), S.Column(copy_atm_to_radiative_grid(S.modern_earth_column(50))), 4)]
for name, tb, col, nzen in cases:
    r = Radtran(tb, len(col["T"]), nzen, 0.15)
    r.upload_column(*col.args())
    for rnd in range(3):
        r.bench_resident_sync(50)
        a = np.median(r.bench_resident_sync(300))
        g = np.median(r.bench_resident_graph(300, 1))
        for _ in range(40): r.radiate_resident()
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(400): r.radiate_resident()
        r.synchronize()
        b = (time.perf_counter() - t0) / 400 * 1e6
        gb = np.median(r.bench_resident_graph(60, 20)) / 20
        print("%s, round %d: call + synchronise %.1f us (graph %.1f); calls back to back %.1f us per call (graph %.1f)" % (name, rnd + 1, a, g, b, gb), flush=True)
