#!/usr/bin/env python3
"""Generate tests/golden/twostream_golden.npz from the REFERENCE two-stream solver.

Runs only in the build container: it calls oracle/_ref/libclima_twostream_ref.so, which
oracle/Makefile compiles with amdflang, unmodified, from
/root/reference/src/clima_const.f90 + src/radtran/clima_radtran_twostream.f90
(two_stream_ir :156-295, two_stream_solar :10-154, tridiag :297-316).  Only inputs and the
reference's outputs are stored (data, no source).

Cases 0-35 cover: nz = 1, 2, 3, 7, 50, 200; optically thin (tau ~ 1e-8, below ir_tau_min) to
thick (tau ~ 1e3, exp underflow); w0 from 0 to the cap 0.99999; g up to the cap 0.999999;
hard / no hard surface; emissivity < 1; grazing and overhead sun; albedo 0 and 1.  Their Planck
values are random over four decades from level to level, which makes the IR source slope
(B_{i+1}-B_i)/tau of thin layers deliberately ill-conditioned.
Cases 36-67: smooth Planck profiles and pressure-like optical depths (the shape of a real call) at
nz = 1 ... 512, including every chunk edge of the 64-lane decomposition.
Cases 68-89: the same kind of column with pairwise identical layers (the doubled radiative grid of
AdiabatClimate), nz = 2 ... 512 even, for the paired kernel instantiations.
Response cases r00...: smooth and paired cases again with a few of their Planck values changed (a column of the RCE
Jacobian against its base profile), for the response form of the batched IR call.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # noqa: E402


def main():
    assert O.ref_available(), "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.default_rng(20260515)
    out = {}
    n = 0
    for nz in (1, 2, 3, 7, 50, 200):
        for variant in range(6):
            if variant == 0:
                tau = 10 ** rng.uniform(-6, 2, nz)
            elif variant == 1:
                tau = 10 ** rng.uniform(-9, -5, nz)          # all thinner than ir_tau_min
            elif variant == 2:
                tau = 10 ** rng.uniform(1, 3, nz)            # thick: exp(-lambda*tau) underflows
            else:
                tau = 10 ** rng.uniform(-8, 2.5, nz)
            w0 = rng.uniform(0, 0.99999, nz)
            g = rng.uniform(0, 0.9, nz)
            if variant == 3:
                w0[:] = 0.99999                               # max_w0 cap (types.f90:9)
                g[:] = 0.999999                               # max_gt cap (types.f90:10)
            if variant == 4:
                w0[:] = 0.0
                g[:] = 0.0
            bp = 10 ** rng.uniform(-13, -9, nz + 1)
            hs = variant % 2 == 0
            em = 1.0 if variant < 2 else float(rng.uniform(0.3, 1.0))
            fup, fdn = O.ref_two_stream_ir(tau, w0, g, em, hs, 1e-6, bp)
            u0 = [0.5, 0.02, 1.0, 0.3, 0.9, 0.7][variant]
            rs = [0.3, 0.0, 1.0, 0.15, 0.5, 0.9][variant]
            am, sr, sfup, sfdn = O.ref_two_stream_solar(tau, w0, g, u0, rs)
            k = "c%02d_" % n
            out.update({k + "tau": tau, k + "w0": w0, k + "g": g, k + "bplanck": bp,
                        k + "ir_par": np.array([em, float(hs), 1e-6]), k + "ir_fup": fup, k + "ir_fdn": fdn,
                        k + "sol_par": np.array([u0, rs]), k + "sol_amean": am, k + "sol_sr": np.array([sr]),
                        k + "sol_fup": sfup, k + "sol_fdn": sfdn})
            n += 1
    # ---- cases 36...: smooth Planck profiles (B_nu of a lapse-rate temperature profile at one
    # frequency, as radiate.f90:63-69 hands them over) and optical depths that grow with pressure --
    # the well-conditioned shape of a real call -- at column heights that reach every slot count of
    # the wave kernels (64 layers per slot: 1...8), including the chunk edges 64/65, 128/129, 256/257.
    rng2 = np.random.default_rng(20261004)
    h, kb, cl = 6.62607004e-34, 1.380649e-23, 299792458.0
    for nz in (1, 5, 50, 64, 65, 128, 129, 192, 200, 256, 257, 300, 402, 448, 500, 512):
        for variant in range(2):
            lev = np.linspace(0.0, 1.0, nz + 1)                      # TOA -> ground
            T = 180.0 + 110.0 * lev ** (1.0 if variant == 0 else 2.5) + rng2.uniform(-1.0, 1.0, nz + 1)
            nu = [2.0e13, 6.0e13][variant]
            bp = 1.0e3 * ((2.0 * h * nu ** 3) / cl ** 2) / (np.exp((h * nu) / (kb * T)) - 1.0)
            mid = 0.5 * (lev[1:] + lev[:-1])
            tau = (10.0 ** rng2.uniform(-3, 1.5)) * (mid ** 2 + 1e-4) / nz * 40.0 * 10 ** rng2.uniform(-0.3, 0.3, nz)
            w0 = rng2.uniform(0.0, 0.9 if variant == 0 else 0.3, nz)
            g = rng2.uniform(0.0, 0.85, nz)
            hs = variant == 0
            em = 1.0 if variant == 0 else 0.9
            fup, fdn = O.ref_two_stream_ir(tau, w0, g, em, hs, 1e-6, bp)
            u0 = [0.6, 0.25][variant]
            rs = [0.2, 0.6][variant]
            am, sr, sfup, sfdn = O.ref_two_stream_solar(tau, w0, g, u0, rs)
            k = "c%02d_" % n
            out.update({k + "tau": tau, k + "w0": w0, k + "g": g, k + "bplanck": bp,
                        k + "ir_par": np.array([em, float(hs), 1e-6]), k + "ir_fup": fup, k + "ir_fdn": fdn,
                        k + "sol_par": np.array([u0, rs]), k + "sol_amean": am, k + "sol_sr": np.array([sr]),
                        k + "sol_fup": sfup, k + "sol_fdn": sfdn})
            n += 1
    # ---- cases 68...: columns of pairwise identical layers (tau, w0, g of layer 2m+1 = those of layer 2m; the
    # levels' Planck values stay their own) -- what AdiabatClimate's doubled radiative grid hands to the solver
    # (src/adiabat/clima_adiabat.f90:729-773), and what the PAIRED instantiations of the fused grid's two-stream
    # part are selected for.  Heights reach every paired slot count (2, 4, 6, 8 = 2 ceil((nz/2)/64)) and its edges.
    out["paired_first"] = np.array([n])
    rng3 = np.random.default_rng(20261006)
    for nz in (2, 66, 102, 128, 130, 202, 256, 258, 384, 402, 512):
        for variant in range(2):
            lev = np.linspace(0.0, 1.0, nz + 1)                      # TOA -> ground
            T = 175.0 + 120.0 * lev ** (1.0 if variant == 0 else 2.0) + rng3.uniform(-1.0, 1.0, nz + 1)
            nu = [3.0e13, 5.0e13][variant]
            bp = 1.0e3 * ((2.0 * h * nu ** 3) / cl ** 2) / (np.exp((h * nu) / (kb * T)) - 1.0)
            nh = nz // 2
            midh = (np.arange(nh) + 0.5) / nh
            tau_h = (10.0 ** rng3.uniform(-3, 1.5)) * (midh ** 2 + 1e-4) / nz * 40.0 * 10 ** rng3.uniform(-0.3, 0.3, nh)
            w0_h = rng3.uniform(0.0, 0.9 if variant == 0 else 0.3, nh)
            g_h = rng3.uniform(0.0, 0.85, nh)
            tau, w0, g = np.repeat(tau_h, 2), np.repeat(w0_h, 2), np.repeat(g_h, 2)
            hs = variant == 0
            em = 1.0 if variant == 0 else 0.85
            fup, fdn = O.ref_two_stream_ir(tau, w0, g, em, hs, 1e-6, bp)
            u0 = [0.55, 0.3][variant]
            rs = [0.25, 0.5][variant]
            am, sr, sfup, sfdn = O.ref_two_stream_solar(tau, w0, g, u0, rs)
            k = "c%02d_" % n
            out.update({k + "tau": tau, k + "w0": w0, k + "g": g, k + "bplanck": bp,
                        k + "ir_par": np.array([em, float(hs), 1e-6]), k + "ir_fup": fup, k + "ir_fdn": fdn,
                        k + "sol_par": np.array([u0, rs]), k + "sol_amean": am, k + "sol_sr": np.array([sr]),
                        k + "sol_fup": sfup, k + "sol_fdn": sfdn})
            n += 1
    out["ncases"] = np.array([n])
    # ---- response cases r00...: a smooth case above with a FEW of its Planck values changed (what a column of the RCE
    # Jacobian is to the base profile, src/adiabat/clima_adiabat_solve.f90:798-812): the reference's two_stream_ir on
    # the changed profile.  Levels next to the top and the surface, neighbours, several per column; changes of 3 %
    # (a few K) and of 1e-4 relative.  tests/test_gpu_golden.py holds F(base) + the response form's changes to these.
    rng4 = np.random.default_rng(20261101)
    m = 0
    for base in range(36, n):
        nz = len(out["c%02d_tau" % base])
        if nz < 4 or (base % 2 == 1 and nz not in (5, 65, 129, 257, 402)):
            continue
        k = "c%02d_" % base
        tau, w0, g, bp, par = (out[k + x] for x in ("tau", "w0", "g", "bplanck", "ir_par"))
        cols = [[0], [1], [nz], [nz - 1], [nz // 2], [0, nz], [nz // 3, nz // 3 + 1], [2, nz // 2, nz - 2],
                sorted(set(int(x) for x in rng4.integers(0, nz + 1, 3)))]
        r = "r%02d_" % m
        out[r + "base"] = np.array([base])
        out[r + "ncol"] = np.array([len(cols)])
        for j, ks in enumerate(cols):
            rel = 1e-4 if j % 3 == 2 else 0.03
            bp2 = bp.copy()
            for kk in ks:
                bp2[kk] = bp[kk] * (1.0 + rel * (1.0 + rng4.random()))
            fup2, fdn2 = O.ref_two_stream_ir(tau, w0, g, float(par[0]), bool(par[1]), float(par[2]), bp2)
            out.update({r + "c%d_k" % j: np.array(ks), r + "c%d_b" % j: bp2[ks], r + "c%d_fup" % j: fup2, r + "c%d_fdn" % j: fdn2})
        m += 1
    out["nresp"] = np.array([m])
    path = os.path.join(HERE, "twostream_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, n, "cases", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
