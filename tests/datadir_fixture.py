"""Writes a data directory in the schema of the reference's opacity data package
(`photochem_clima_data`: kdistributions/, CIA/, xsections/, water_continuum/, rayleigh/,
aerosol_xsections/), a settings YAML and a stellar spectrum from a synthetic TableSet, the way
the reference's loader expects to find them (src/radtran/clima_radtran_types_create.f90).
Test infrastructure: the real package is a network fetch and is not in this image.

Cross sections that the loader regrids are written on their own, finer wavelength grids.
"""
import os

import numpy as np

from clima_amd import h5lite
from clima_amd import synthetic as S

RAY_PAR = {"CO2": (43.9e-5, 6.4e-3, 0.0805), "O2": (26.63e-5, 5.07e-3, 0.054), "N2": (29.06e-5, 7.7e-3, 0.0305),
           "CH4": (42.6e-5, 14.41e-3, 0.0), "H2O": (28.0e-5, 5.0e-3, 0.17)}


def write_datadir(root, tb, rng=None, fine=3, h5write=None):
    """`tb`: TableSet from clima_amd.synthetic.make_tables.  Returns the dict of what was written
    (fine-grid inputs of the regridded quantities) for the tests to check against.
    `h5write(path, {name: array})` replaces clima_amd.h5lite.write (tests/golden/make_datadir_c.py
    writes the HDF5 files through the HDF5 C library's chunk / deflate / float32 path instead)."""
    rng = rng or np.random.default_rng(3)
    h5w = h5write or h5lite.write
    sp = list(tb.species_names)
    for d in ("kdistributions", "CIA", "xsections", "water_continuum", "rayleigh", "aerosol_xsections"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    wavl_um = tb.wavl / 1.0e3
    for k in tb.ktables:
        h5w(os.path.join(root, "kdistributions", sp[k["sp_ind"]] + ".h5"),
                     {"weights": k["weights"], "log10P": k["log10P"], "T": k["temp"], "wavelengths": wavl_um,
                      "log10k": k["log10k"]})          # C (nwav, ntemp, npress, ngauss)
    h5w(os.path.join(root, "kdistributions", "bins.h5"),
                 {"ir_wavl": tb.ir_wavl / 1.0e3, "sol_wavl": tb.sol_wavl / 1.0e3})
    written = {"cia": {}, "pxs": {}, "cont": None, "part": {}}
    # point grid for the regridded quantities: `fine` points per bin, inside the opacity grid only
    xf = np.unique(np.concatenate([np.geomspace(tb.wavl[i], tb.wavl[i + 1], fine + 1) for i in range(tb.nw)]))[2:-2]
    lx = np.log10(xf)
    for x in tb.xsections:
        if x["xs_type"] == S.XS_CIA:
            name = sp[x["sp1"]] + "-" + sp[x["sp2"]]
            temp = x["temp"]
            vals = -46.0 + 1.5 * np.sin(3.0 * lx)[:, None] + 0.003 * (temp - 300.0)[None, :] + \
                rng.uniform(-0.2, 0.2, (len(xf), len(temp)))
            h5w(os.path.join(root, "CIA", name + ".h5"),
                         {"wavelengths": xf / 1.0e3, "T": temp, "log10xs": vals})   # C (nwav, ntemp)
            written["cia"][name] = (xf, temp, vals)
        elif x["xs_type"] == S.XS_PHOTOLYSIS:
            name = sp[x["sp1"]]
            m = xf < 400.0
            xs = 1.0e-18 * np.exp(-((xf[m] / 200.0) ** 4)) * 10.0 ** rng.uniform(-0.3, 0.3, m.sum())
            h5w(os.path.join(root, "xsections", name + ".h5"), {"wavelengths": xf[m], "photoabsorption": xs})
            written["pxs"][name] = (xf[m], xs)
    ray = [sp[x["sp1"]] for x in tb.xsections if x["xs_type"] == S.XS_RAYLEIGH]
    with open(os.path.join(root, "rayleigh", "rayleigh.yaml"), "w") as f:
        for name, (A, B, D) in RAY_PAR.items():
            f.write("%s:\n  formalism: vardavas\n  data: {A: %r, B: %r, Delta: %r}\n" % (name, A, B, D))
    if tb.continuum is not None:
        temp = tb.continuum["temp"]
        a = -44.0 - 0.5 * (lx - 2.0)[:, None] - 0.005 * (temp - 296.0)[None, :] + rng.uniform(-0.1, 0.1, (len(xf), len(temp)))
        b = a - 2.0 + rng.uniform(-0.1, 0.1, a.shape)
        h5w(os.path.join(root, "water_continuum", "MT_CKD.h5"),
                     {"wavelengths": xf / 1.0e3, "T": temp, "log10xs_H2O": a, "log10xs_foreign": b})
        written["cont"] = (xf, temp, a, b)
    for p_, pname in zip(tb.particles, tb.particle_names):
        rad_um = p_["radii"] * 1.0e4
        size = 2.0 * np.pi * p_["radii"][None, :] / (xf[:, None] * 1.0e-7)
        qext = 2.0 * size ** 4 / (1.0 + size ** 4) + 1.0e-12
        w0 = np.clip(0.2 + 0.75 * size ** 2 / (1.0 + size ** 2), 0.0, 0.999)
        g0 = 0.8 * size ** 2 / (1.0 + size ** 2)
        os.makedirs(os.path.join(root, "aerosol_xsections", "khare1984"), exist_ok=True)
        h5w(os.path.join(root, "aerosol_xsections", "khare1984", "mie_khare1984.h5"),
                     {"wavelengths": xf, "radii": rad_um, "w0": w0, "qext": qext, "g0": g0})   # C (nwav, nrad)
        written["part"][pname] = (xf, rad_um, w0, qext, g0)
    kd = [sp[k["sp_ind"]] for k in tb.ktables]
    with open(os.path.join(root, "settings.yaml"), "w") as f:
        f.write("atmosphere-grid:\n  bottom: 0.0\n  top: 1.0e7\n  number-of-layers: 50\n\n")
        f.write("optical-properties:\n  species:\n    gases: [%s]\n    particles: [%s]\n" % (", ".join(sp), ", ".join(tb.particle_names)))
        f.write("  k-method: RandomOverlapResortRebin\n")
        parts = "[" + ", ".join("{name: %s, data: khare1984}" % n for n in tb.particle_names) + "]"
        f.write("  opacities: {k-distributions: [%s], CIA: true, rayleigh: [%s], photolysis-xs: true,\n"
                "    water-continuum: MT_CKD%s}\n" % (", ".join(kd), ", ".join(ray),
                                                      (", particle-xs: " + parts) if tb.particle_names else ""))
    # stellar spectrum: wavelength nm, flux mW/m^2/nm, one header line
    ws = np.geomspace(tb.wavl[0] * 0.9, 1.0e5, 4000)
    nu = 2.99792458e8 / (ws * 1.0e-9)
    xx = np.minimum(6.62607004e-34 * nu / (1.380649e-23 * 5772.0), 700.0)
    flux = 2.16e-5 * np.pi * 1.0e3 * 2.0 * 6.62607004e-34 * nu ** 3 / 2.99792458e8 ** 2 / np.expm1(xx) * nu / ws
    with open(os.path.join(root, "star.txt"), "w") as f:
        f.write("Wavelength (nm)      Solar flux (mW/m^2/nm)\n")
        for a, b in zip(ws, flux):
            f.write("%.10e   %.10e\n" % (a, b))
    written["star"] = (ws, flux)
    return written
