#!/usr/bin/env python3
"""Derive the compact numeric input fixtures under clima_amd/data/ from the reference's
template inputs (run in the build container only; /root/reference is absent on the GPU box).

Outputs (numeric data only, no reference source text):
  clima_amd/data/modern_earth_atmosphere.npz
      alt_km, press_bar, temp_K, species (names), mix (nrow, nspecies)
      <- templates/ModernEarth/atmosphere.txt (200 rows)
  clima_amd/data/stellar_binned.npz
      wavl_nm (1001 edges of the nominal synthetic grid), sun_now, sun_3p8Ga
      (mW/m^2/Hz per bin, all 1000 bins) <- templates/ModernEarth/Sun_now.txt and
      templates/AdiabatClimate/Mars/Sun_3.8Ga.txt binned with the recipe of
      read_stellar_flux (src/radtran/clima_radtran_types_create.f90:9-78): pad the point
      spectrum with zero-flux points just outside its range (addpnt), average the
      piecewise-linear spectrum over each bin (inter2), convert mW/m^2/nm -> mW/m^2/Hz
      with the bin-mean wavelength (:70-76).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clima_amd.synthetic import nominal_wavl  # noqa: E402

REF = os.environ.get("CLIMA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "clima_amd", "data")
C_LIGHT = 299792458.0


def bin_average(edges, x, y):
    """inter2 semantics: mean of the piecewise-linear (x,y) over each [edges[i],edges[i+1]]."""
    # cumulative integral of the piecewise-linear function at the data nodes
    F = np.concatenate([[0.0], np.cumsum(0.5 * (y[1:] + y[:-1]) * np.diff(x))])

    def Fat(t):
        i = np.clip(np.searchsorted(x, t, side="right") - 1, 0, len(x) - 2)
        dx = t - x[i]
        slope = (y[i + 1] - y[i]) / (x[i + 1] - x[i])
        return F[i] + y[i] * dx + 0.5 * slope * dx * dx

    return (Fat(edges[1:]) - Fat(edges[:-1])) / np.diff(edges)


def read_stellar_flux(path, wavl):
    d = np.loadtxt(path, skiprows=1)
    w, f = d[:, 0], d[:, 1]
    rdelta = 1.0e-4
    # addpnt x4 (types_create.f90:54-57)
    w = np.concatenate([[0.0, w[0] * (1.0 - rdelta)], w, [w[-1] * (1.0 + rdelta), 1.0e300]])
    f = np.concatenate([[0.0, 0.0], f, [0.0, 0.0]])
    flux = bin_average(wavl, w, f)  # mW/m2/nm
    wavl_av = 0.5 * (wavl[:-1] + wavl[1:])
    return flux * (((wavl_av * 1.0e-9) * wavl_av) / C_LIGHT)  # mW/m2/Hz (:73-76)


def main():
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(REF, "templates/ModernEarth/atmosphere.txt")
    with open(path) as fh:
        labels = fh.readline().split()
    d = np.loadtxt(path, skiprows=1)
    species = [s for s in labels if s not in ("alt", "press", "temp")]
    np.savez_compressed(
        os.path.join(OUT, "modern_earth_atmosphere.npz"),
        alt_km=d[:, labels.index("alt")], press_bar=d[:, labels.index("press")],
        temp_K=d[:, labels.index("temp")], species=np.array(species),
        mix=np.stack([d[:, labels.index(s)] for s in species], axis=1))
    wavl = nominal_wavl()
    np.savez_compressed(
        os.path.join(OUT, "stellar_binned.npz"), wavl_nm=wavl,
        sun_now=read_stellar_flux(os.path.join(REF, "templates/ModernEarth/Sun_now.txt"), wavl),
        sun_3p8Ga=read_stellar_flux(os.path.join(REF, "templates/AdiabatClimate/Mars/Sun_3.8Ga.txt"), wavl))
    print("wrote fixtures to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
