"""Columns with many changed levels: response form against the general kernel (accuracy, time)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from clima_amd import synthetic as S
from clima_amd.atmosphere import copy_atm_to_radiative_grid
from clima_amd.radtran import Radtran
tb = S.modern_earth_tables()
for nz, ndev in ((200, 12), (200, 20), (200, 25), (100, 4), (100, 8), (100, 12), (50, 4), (50, 6)):
    col = S.Column(copy_atm_to_radiative_grid(S.modern_earth_column(nz)))
    nzr = len(col["T"])
    r = Radtran(tb, nzr, 4, 0.15)
    r.radiate(*col.args())
    ncol = nzr + 1
    rng = np.random.default_rng(nz + ndev)
    T = np.repeat(np.asarray(col["T"])[:, None], ncol, axis=1)
    Ts = np.full(ncol, float(col["T_surface"]))
    for c in range(ncol):
        for j in rng.choice(nzr, ndev, replace=False):
            T[j, c] += rng.uniform(-2, 2)
    res = {}
    for mode in (0, 1, 2):
        r.ir_green = mode
        n0 = r.ir_green_batches
        out = r.radiate_ir_batch(Ts, T)
        best = 1e9
        for rep in range(4):
            t0 = time.time(); r.radiate_ir_batch(Ts, T, out=out, pin=True); best = min(best, time.time() - t0)
        res[mode] = (best, [np.array(x) for x in out], r.ir_green_batches > n0)
    dev = max(float(np.max(np.abs(a - b)) / np.max(np.abs(b))) for a, b in zip(res[1][1], res[0][1]))
    print("%d layers, %d columns x %d changed levels each: automatic %.2f ms%s, forced response form %.2f ms%s, general kernel %.2f ms, difference %.1e of the maximum" % (nzr, ncol, ndev, res[1][0] * 1e3, " [response form]" if res[1][2] else " [general]", res[2][0] * 1e3, "" if res[2][2] else " [not taken]", res[0][0] * 1e3, dev), flush=True)
    r.spectra_release()
