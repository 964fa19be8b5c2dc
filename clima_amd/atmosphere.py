"""Caller-side glue in front of `Radtran%radiate`: the reference's atmosphere-file reader,
vertical grid and radiative-grid doubling (SURVEY.md 8(f) "next #2"), so a caller shaped like
tests/test_radtran.f90 or AdiabatClimate can drive the HIP path from the same inputs.

  AtmosphereFile            src/clima_types_create.f90:356-423  (create_AtmosphereFile)
  unpack_atmospherefile     src/clima_types_create.f90:425-513
  vertical_grid             src/clima_eqns.f90:172-184
  column_from_atmosphere    tests/test_radtran.f90:27-67 (densities, dummy particles)
  copy_atm_to_radiative_grid src/adiabat/clima_adiabat.f90:728-771

Error texts follow the reference (they are API there).  The interpolation used by
`unpack_atmospherefile` is futils' `interp(ng, n, xg, x, y, yg, ierr)` (v0.1.14, third-party,
absent from the reference tree): piecewise linear, constant beyond the ends.
"""
import numpy as np

from .radtran import ClimaException

K_BOLTZ = 1.380649e-16  # erg/K, src/clima_const.f90:12


class AtmosphereFile:
    """Whitespace table: one header line of labels, then `nz` rows (alt [km], press [bar],
    temp [K], mixing ratios...).  `columns` is (nlabels, nz) as in the reference."""

    def __init__(self, atm_file):
        self.filename = atm_file
        try:
            with open(atm_file) as f:
                lines = [ln for ln in f.read().splitlines()]
        except OSError:
            raise ClimaException("Can not open file " + atm_file)
        lines = [ln for ln in lines if ln.strip()]
        if not lines:
            raise ClimaException("Can not open file " + atm_file)
        self.labels = lines[0].split()
        rows = lines[1:]
        if rows and len(rows[0].split()) != len(self.labels):
            raise ClimaException("There is a missing column label in the file " + atm_file)
        self.nlabels = len(self.labels)
        self.nz = len(rows)
        cols = np.empty((self.nlabels, self.nz))
        for i, ln in enumerate(rows):
            parts = ln.split()
            try:
                if len(parts) < self.nlabels:
                    raise ValueError
                cols[:, i] = [float(x.replace("d", "e").replace("D", "e")) for x in parts[: self.nlabels]]
            except ValueError:
                raise ClimaException('Problem reading in initial atmosphere in "' + atm_file + '"')
        self.columns = cols


def vertical_grid(bottom, top, nz):
    """z (layer centres) and dz, cm (src/clima_eqns.f90:172-184)."""
    dz = np.full(nz, (top - bottom) / nz)
    z = np.empty(nz)
    z[0] = dz[0] / 2.0
    for i in range(1, nz):
        z[i] = z[i - 1] + dz[i]
    return z, dz


def _interp(xg, x, y, filename):
    if np.any(np.diff(x) <= 0.0):
        raise ClimaException('Error interpolating "' + filename + '"')
    return np.interp(xg, x, y)  # linear inside, constant outside


def unpack_atmospherefile(atm, species_names, z):
    """-> mix (nz, ng), T (nz) K, P (nz) bar at the layer centres `z` [cm]."""
    fn = atm.filename
    if "alt" not in atm.labels:
        raise ClimaException('"alt" was not found in input file "' + fn + '"')
    alt = atm.columns[atm.labels.index("alt")] * 1.0e5
    nz = len(z)
    mix = np.empty((nz, len(species_names)), order="F")
    for i, sp in enumerate(species_names):
        if sp not in atm.labels:
            raise ClimaException('Species "' + sp + '" was not found in "' + fn + '"')
        mix[:, i] = _interp(z, alt, np.log10(atm.columns[atm.labels.index(sp)]), fn)
    mix = 10.0 ** mix
    s = mix.sum(axis=1)
    if np.any(np.abs(s - 1.0) > 1.0e-2 * np.maximum(np.abs(s), 1.0)):
        raise ClimaException('mixing ratios do not sum to close to 1 in "' + fn + '"')
    if "temp" not in atm.labels:
        raise ClimaException('"temp" was not found in input file "' + fn + '"')
    T = _interp(z, alt, atm.columns[atm.labels.index("temp")], fn)
    if "press" not in atm.labels:
        raise ClimaException('"press" was not found in input file "' + fn + '"')
    P = 10.0 ** _interp(z, alt, np.log10(atm.columns[atm.labels.index("press")]), fn)
    return mix, T, P


def column_from_atmosphere(atm, species_names, nz=None, bottom=0.0, top=1.0e7, n_particles=0,
                           T_shift=0.0):
    """The column tests/test_radtran.f90:27-67 builds: uniform grid, unpacked profile,
    densities = mix*P*1e6/(k T), placeholder particles (pdensities 1, radii 1e-5).
    Returns the argument dict of `Radtran.radiate`."""
    if isinstance(atm, str):
        atm = AtmosphereFile(atm)
    nz = atm.nz if nz is None else nz
    z, dz = vertical_grid(bottom, top, nz)
    mix, T, P = unpack_atmospherefile(atm, list(species_names), z)
    T = T + T_shift
    density = (P * 1.0e6) / (K_BOLTZ * T)
    col = dict(T_surface=float(T[0]), T=T, P=P, densities=np.asfortranarray(mix * density[:, None]), dz=dz)
    if n_particles > 0:
        col["pdensities"] = np.asfortranarray(np.full((nz, n_particles), 1.0))
        col["radii"] = np.asfortranarray(np.full((nz, n_particles), 1.0e-5))
    return col


def copy_atm_to_radiative_grid(col, double_radiative_grid=True):
    """AdiabatClimate's radiative grid (src/adiabat/clima_adiabat.f90:728-771): with
    `double_radiative_grid` every layer becomes two half-thickness copies and two ghost
    layers repeat the top one, nz_r = 2*nz + 2 -- pairs that `pair_reuse`
    (clima_radtran_types.f90:621-632) interpolates once."""
    if not double_radiative_grid:
        return dict(col)

    def dbl(a):
        a = np.repeat(np.asarray(a), 2, axis=0)
        return np.concatenate([a, a[-1:], a[-1:]], axis=0)

    out = dict(T_surface=col["T_surface"], T=dbl(col["T"]), P=dbl(col["P"]),
               densities=np.asfortranarray(dbl(col["densities"])))
    half = np.repeat(0.5 * np.asarray(col["dz"]), 2)
    out["dz"] = np.concatenate([half, half[-1:], half[-1:]])
    if col.get("radii") is not None:
        out["pdensities"] = np.asfortranarray(dbl(col["pdensities"]))
        out["radii"] = np.asfortranarray(dbl(col["radii"]))
    return out
