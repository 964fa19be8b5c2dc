#!/usr/bin/env python3
"""Generate tests/golden/twostream_golden.npz from the REFERENCE two-stream solver.

Runs only in the build container: it calls oracle/_ref/libclima_twostream_ref.so, which
oracle/Makefile compiles with amdflang, unmodified, from
/root/reference/src/clima_const.f90 + src/radtran/clima_radtran_twostream.f90
(two_stream_ir :156-295, two_stream_solar :10-154, tridiag :297-316).  Only inputs and the
reference's outputs are stored (data, no source).

Cases cover: nz = 1, 2, 3, 7, 50, 200; optically thin (tau ~ 1e-8, below ir_tau_min) to
thick (tau ~ 1e3, exp underflow); w0 from 0 to the cap 0.99999; g up to the cap 0.999999;
hard / no hard surface; emissivity < 1; grazing and overhead sun; albedo 0 and 1.
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
    out["ncases"] = np.array([n])
    path = os.path.join(HERE, "twostream_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, n, "cases", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
