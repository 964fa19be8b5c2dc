"""Write a TableSet + column as the stream file clima_amd/fortran/radtran_driver.f90 reads."""
import numpy as np


def write_case(path, tables, col, nzen, albedo):
    t = tables
    nz = len(col["T"])
    with open(path, "wb") as f:
        def i32(*v):
            f.write(np.asarray(v, dtype="<i4").tobytes())

        def f64(a):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())

        i32(nz, t.nsp, t.np_, t.nw, nzen)
        f64([albedo])
        f64(t.wavl)
        i32(len(t.ktables))
        for k in t.ktables:
            i32(k["sp_ind"] + 1, len(k["weights"]), len(k["log10P"]), len(k["temp"]))
            f64(k["weights"]); f64(k["log10P"]); f64(k["temp"]); f64(k["log10k"])  # C [nw][nT][nP][ng] == F (ng,nP,nT,nw)
        i32(len(t.xsections))
        for x in t.xsections:
            temp = x.get("temp")
            i32(x["xs_type"], x["dim"], x["sp1"] + 1, x.get("sp2", -1) + 1, 0 if temp is None else len(temp))
            if x["dim"] == 1:
                f64(temp)
            f64(x["data"])  # C [nw][nT] == F (nT,nw)
        i32(0 if t.continuum is None else 1)
        if t.continuum is not None:
            c = t.continuum
            i32(c["LH2O"] + 1, len(c["temp"]))
            f64(c["temp"]); f64(c["log10_H2O"]); f64(c["log10_foreign"])
        i32(len(t.particles))
        for p in t.particles:
            i32(p["p_ind"] + 1, len(p["radii"]))
            f64(p["radii"]); f64(p["w0"]); f64(p["qext"]); f64(p["gt"])
        i32(len(t.ir_wavl)); f64(t.ir_wavl)
        i32(len(t.sol_wavl)); f64(t.sol_wavl)
        f64(t.photons_sol)
        f64([col["T_surface"]]); f64(col["T"]); f64(col["P"])
        f64(np.asfortranarray(col["densities"]).T)  # F (nz,nsp) column-major
        f64(col["dz"])
        if t.np_ > 0:
            f64(np.asfortranarray(col["pdensities"]).T)
            f64(np.asfortranarray(col["radii"]).T)
