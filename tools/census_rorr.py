"""Census of the random-overlap mixing steps (k_rorr, /root/reference/src/radtran/clima_radtran_types.f90:823-852):
how often are the ng*ng sums x_i + y_j of a step already in ascending order as the reference lays them out
(row-major: range(y) <= smallest gap of x) or in the transposed order (range(x) <= smallest gap of y) -- and how
often does that hold for EVERY lane of a 64-lane wave of the opacity tile (lane = (bin, source layer), consecutive
source layers of a bin; clima_amd/csrc/kernels.hip opacity8_body), which is what a wave-uniform fast path needs.

CPU only: runs the oracle with its census hook.   python tools/census_rorr.py > profiles/r04_census_rorr.txt
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clima_amd import synthetic as S  # noqa: E402
from clima_amd.atmosphere import copy_atm_to_radiative_grid  # noqa: E402
from oracle import oracle as O  # noqa: E402


def census(tables, col, nz, nzen, albedo):
    ref = O.OracleRadtran(tables, nz, nzen, albedo)
    L = O.lib()
    L.orc_census_set.argtypes = [C.c_void_p]
    nw, nk = tables.nw, len(tables.ktables)
    buf = np.zeros((nw, nk - 1, nz, 4), dtype=np.uint8)
    L.orc_census_set(buf.ctypes.data)
    try:
        ref.radiate(*col.args(), compute_solar=False)
    finally:
        L.orc_census_set(None)
    return buf


def report(name, buf):
    nw, ns, nz, _ = buf.shape
    f = buf[..., 0]
    src = (f[0, 0] & 64) == 0                       # source layers (same for every bin and step)
    fl = f[:, :, src]                               # [bin][step][source layer]
    nsrc = fl.shape[2]
    items = fl.size
    print("== %s: %d bins x %d steps x %d source layers (of %d) = %d mixing steps" % (name, nw, ns, nsrc, nz, items))
    for bit, what in ((1, "y ascending"), (2, "x ascending"), (4, "row-major (range y <= min gap x)"),
                      (8, "column-major (range x <= min gap y)"), (16, "row-major, relaxed by 2^-46 of the smallest sum"),
                      (32, "column-major, relaxed")):
        print("   %-52s %7.3f %% of steps" % (what, 100.0 * np.count_nonzero(fl & bit) / items))
    either = (fl & 12) != 0
    either_r = (fl & 48) != 0
    print("   %-52s %7.3f %%   relaxed %7.3f %%" % ("either order known", 100.0 * either.mean(), 100.0 * either_r.mean()))
    for s in range(ns):
        print("      step %d (species %d onto the mixture): row %6.2f %%  column %6.2f %%  either %6.2f %%" %
              (s + 1, s + 2, 100.0 * ((fl[:, s] & 4) != 0).mean(), 100.0 * ((fl[:, s] & 8) != 0).mean(), 100.0 * either[:, s].mean()))
    # waves: lane t = bin * nsrc + k, 64 consecutive t (kernels.hip: t = tile*256 + tid, l = t / nsrc, source t % nsrc)
    for label, e in (("strict", fl), ("relaxed", fl >> 2)):
        tot = 0
        hit = {"row": 0, "col": 0, "mixed(lanewise either)": 0}
        for s in range(ns):
            a = e[:, s, :].reshape(-1)
            nwave = (a.size + 63) // 64
            pad = np.full(nwave * 64 - a.size, 255, dtype=np.uint8)
            w = np.concatenate([a, pad]).reshape(nwave, 64)
            row = ((w & 4) != 0).all(axis=1)
            colm = ((w & 8) != 0).all(axis=1)
            mix = ((w & 12) != 0).all(axis=1)
            tot += nwave
            hit["row"] += int(row.sum())
            hit["col"] += int((colm & ~row).sum())
            hit["mixed(lanewise either)"] += int((mix & ~row & ~colm).sum())
        print("   waves (%s): %d (wave, step) pairs; whole wave row-major %.3f %%, column-major %.3f %%, every lane one of the two %.3f %%"
              % (label, tot, 100.0 * hit["row"] / tot, 100.0 * hit["col"] / tot, 100.0 * hit["mixed(lanewise either)"] / tot))
    nx = buf[..., 1][:, :, src]
    ny = buf[..., 2][:, :, src]
    inv = buf[..., 3][:, :, src]
    print("   neighbouring rows that interleave (x gaps < range y): histogram 0..7:", np.bincount(nx.reshape(-1), minlength=8)[:8].tolist())
    print("   neighbouring columns that interleave:                 histogram 0..7:", np.bincount(ny.reshape(-1), minlength=8)[:8].tolist())
    q = np.percentile(inv.reshape(-1), [10, 50, 90])
    print("   inversions of the row-major order against the sorted one (of 2016, saturated at 255): p10 %d p50 %d p90 %d" % tuple(q))


def main():
    me = S.modern_earth_tables()
    report("config 2 (ModernEarth, 200 layers)", census(me, S.modern_earth_column(200), 200, 8, 0.15))
    mars = S.early_mars_tables()
    report("config 3 (EarlyMars, 200 layers)", census(mars, S.early_mars_column(200), 200, 4, 0.2))
    report("config 5 (ModernEarth, 500 layers)", census(me, S.modern_earth_column(500), 500, 8, 0.15))
    for nzb in (50, 100, 200):
        colr = S.Column(copy_atm_to_radiative_grid(S.modern_earth_column(nzb)))
        nzr = 2 * nzb + 2
        report("AdiabatClimate doubled grid, %d layers" % nzr, census(me, colr, nzr, 8, 0.15))


if __name__ == "__main__":
    main()
