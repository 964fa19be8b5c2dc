#!/usr/bin/env python3
"""Developer diagnostic (GPU box): HIP path vs oracle on a few cases, with per-kernel times."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from clima_amd import synthetic as S  # noqa: E402
from clima_amd.radtran import Radtran  # noqa: E402
from oracle import oracle as O  # noqa: E402


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    den = np.maximum(np.abs(b), 1e-300)
    return float(np.max(np.abs(a - b) / den)) if a.size else 0.0


def scaled_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def run(name, tables, col, nz, nzen, albedo, reps=5, **scalars):
    print("==", name, "nz", nz, "nw", tables.nw, "nzen", nzen, flush=True)
    r = Radtran(tables, nz, nzen, albedo)
    o = O.OracleRadtran(tables, nz, nzen, albedo)
    for k, v in scalars.items():
        setattr(r, k, v)
    if scalars:
        o.set_scalars(**scalars)
    t0 = time.time()
    isr_o, olr_o = o.TOA_fluxes(*col.args())
    t_cpu = time.time() - t0
    isr, olr = r.TOA_fluxes(*col.args())
    print("  ISR %.6f / %.6f   OLR %.6f / %.6f (W/m2)   rel %.2e %.2e" % (
        isr / 1e3, isr_o / 1e3, olr / 1e3, olr_o / 1e3, abs(isr - isr_o) / abs(isr_o), abs(olr - olr_o) / abs(olr_o)))
    tau, w0, g, tb = r.opr()
    tau_o, w0_o, g_o, tb_o = o.opr()
    print("  opr: tau %.2e  w0 %.2e  g %.2e  tau_band %.2e (max rel)" % (
        relerr(tau, tau_o), relerr(w0, w0_o), relerr(g, g_o), relerr(tb, tb_o)))
    for nm, wg, wo in (("ir", r.wrk_ir, o.wrk_ir), ("sol", r.wrk_sol, o.wrk_sol)):
        print("  %s: fup_n %.2e fdn_n %.2e | fup_a %.2e fdn_a %.2e amean %.2e tau_band %.2e (scaled)" % (
            nm, scaled_err(wg.fup_n, wo.fup_n), scaled_err(wg.fdn_n, wo.fdn_n), scaled_err(wg.fup_a, wo.fup_a),
            scaled_err(wg.fdn_a, wo.fdn_a), scaled_err(wg.amean, wo.amean), scaled_err(wg.tau_band, wo.tau_band)))
    print("  f_total %.2e" % scaled_err(r.f_total, o.f_total))
    # timing, resident
    r.upload_column(*col.args())
    r.profile(True)
    r.profile_reset()
    for _ in range(3):
        r.radiate_resident()
    r.synchronize()
    r.profile_reset()
    t0 = time.time()
    for _ in range(reps):
        r.radiate_resident()
    r.synchronize()
    dt = (time.time() - t0) / reps
    names = ["prep", "opacity", "twostream", "integrate"]
    ks = [r.kernel_time(i) for i in range(4)]
    print("  GPU %.1f us/call (%.0f calls/s)  CPU oracle %.3f s | kernels(us): %s" % (
        dt * 1e6, 1.0 / dt, t_cpu, ", ".join("%s %.1f" % (n, 1e3 * ms / max(c, 1)) for n, (ms, c) in zip(names, ks))))
    r.profile(False)
    t0 = time.time()
    for _ in range(reps):
        r.radiate_resident()
    r.synchronize()
    dt = (time.time() - t0) / reps
    print("  GPU (events off) %.1f us/call (%.0f calls/s)" % (dt * 1e6, 1.0 / dt))
    # PCIe-inclusive: host arrays in, ISR/OLR out through radtran_toa_fluxes_wrapper, synchronous
    args = col.args()
    t0 = time.time()
    for _ in range(reps):
        r.TOA_fluxes(*args)
    dt = (time.time() - t0) / reps
    print("  host API (PCIe-inclusive, synchronous TOA_fluxes) %.1f us/call (%.0f calls/s)" % (dt * 1e6, 1.0 / dt))
    print("  bytes:", r.algorithmic_bytes())


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "small"):
        tb = S.modern_earth_tables(nw=40)
        run("small ModernEarth", tb, S.modern_earth_column(50), 50, 4, 0.3)
        run("small no hard surface", tb, S.modern_earth_column(50), 50, 1, 0.3, has_hard_surface=False)
        run("small doubled (pair_reuse)", tb, S.doubled_column(S.modern_earth_column(25)), 50, 2, 0.2)
        tbu = S.modern_earth_tables(nw=24, sorted_k=False, seed=11)
        run("unsorted k", tbu, S.modern_earth_column(31), 31, 3, 0.1)
    if which in ("all", "nominal"):
        tb = S.modern_earth_tables()
        run("config 2: ModernEarth nominal", tb, S.modern_earth_column(200), 200, 8, 0.15, reps=20)
    if which in ("jacobian",):
        # the RCE Jacobian's radiative work: nz+1 IR-only calls on shared opacities, one at a
        # time through the host API vs one batched call
        tb = S.modern_earth_tables()
        nz = 200
        col = S.modern_earth_column(nz)
        r = Radtran(tb, nz, 8, 0.15)
        r.TOA_fluxes(*col.args())
        n = nz + 1
        T = np.repeat(np.asarray(col["T"])[:, None], n, axis=1)
        Ts = np.full(n, float(col["T_surface"]))
        Ts[0] *= 1.01
        for c in range(1, n):
            T[c - 1, c] *= 1.01
        r.radiate_ir_batch(Ts, T)
        t0 = time.time()
        for _ in range(3):
            fup, fdn, ft = r.radiate_ir_batch(Ts, T)
        tb_ = (time.time() - t0) / 3
        a = list(col.args())
        t0 = time.time()
        ref = np.empty_like(ft)
        for c in range(n):
            a[0] = Ts[c]; a[1] = np.ascontiguousarray(T[:, c])
            r.radiate(*a, compute_solar=False, compute_opacity=False)
            ref[:, c] = r.f_total
        tl = time.time() - t0
        print("== Jacobian radiative work, %d IR-only calls on shared opacities (nz %d)" % (n, nz))
        print("  one at a time %.2f ms (%.1f us/call)   batched %.2f ms (%.1f us/column)   speedup %.1fx   max |diff| f_total %.2e"
              % (tl * 1e3, tl / n * 1e6, tb_ * 1e3, tb_ / n * 1e6, tl / tb_, float(np.max(np.abs(ref - ft)))))
    if which in ("all", "mars"):
        tb = S.early_mars_tables()
        run("config 3: EarlyMars", tb, S.early_mars_column(200), 200, 4, 0.2, reps=20, photon_scale_factor=0.4286)


if __name__ == "__main__":
    main()
