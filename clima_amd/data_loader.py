"""Opacity-data loader: from a `photochem_clima_data`-style data directory + a settings YAML +
a stellar spectrum to the tables `Radtran` consumes (SURVEY.md 8(f) "next #1").

It follows the reference's loader, src/radtran/clima_radtran_types_create.f90:

  read_stellar_flux            :9-78      star file -> photons_sol per solar bin
  create_RTChannel / read_wavl :226-270, :647-687   kdistributions/bins.h5 (ir_wavl, sol_wavl [um])
  create_OpticalProperties     :272-645   which opacities exist / are asked for, file layout
  create_Ktable                :1265-1378 kdistributions/{sp}.h5 (weights, log10P, T, wavelengths, log10k)
  read_h5_Xsection             :1105-1263 CIA/{A}-{B}.h5 (wavelengths, [T,] log10xs) regridded to the bins
  create_WaterContinuum        :868-1046  water_continuum/{model}.h5
  create_RayleighXsection      :1048-1088 rayleigh/rayleigh.yaml (A, B, Delta)
  create_PhotolysisXsection    :1407-1468 xsections/{sp}.h5 (wavelengths, photoabsorption)
  create_ParticleXsection      :734-866   aerosol_xsections/{dat}/mie_{dat}.h5
  settings                     src/clima_types_create.f90:578-600, :737-1000 (optical-properties)

and `Radtran.from_files(...)` below has the argument list of the reference constructor
`Radtran(settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir, err)`
(src/radtran/clima_radtran.f90:98-126).  Error texts are the reference's.

Status.  The opacity data package itself is not in this image (a network fetch in the
reference's build), so this loader is exercised against data directories written in the same
schema by tests/datadir_fixture.py, not against the real files.  Dataset names, units and
array orientation are taken from the Fortran reader (HDF5 presents a Fortran array
`a(n1,n2,...)` to C with the dimensions reversed and the same bytes).  Regridding uses the
third-party futils v0.1.14 routines `addpnt`, `inter2`, `interp_discrete_to_bins`, restated
here from their published behaviour -- parity unpinned (DESIGN.md "Oracle").
"""
import os

import numpy as np

from . import h5lite
from .radtran import ClimaException

C_LIGHT = 299792458.0           # src/clima_const.f90
LOG10TINY = float(np.log10(np.sqrt(np.finfo(np.float64).tiny)))   # src/clima_const.f90:21: log10(sqrt(tiny(1.0_dp)))
HUGE = np.finfo(np.float64).max
RDELTA = 1.0e-4

XS_CIA, XS_RAYLEIGH, XS_ABSORPTION, XS_PHOTOLYSIS = 0, 1, 2, 3


def _pow10(a):
    """10**a through the C library's pow, element by element: numpy's vectorised power / log10 may be SIMD routines
    of their own that differ from libm in the last bit (and from host to host), and the loader behind the C ABI
    (csrc/radtran_loader.hip) has to produce the same tables bit for bit."""
    import math
    return np.array([math.pow(10.0, float(v)) for v in np.asarray(a, dtype=float).ravel()]).reshape(np.shape(a))


def _log10(a):
    import math
    return np.array([math.log10(float(v)) for v in np.asarray(a, dtype=float).ravel()]).reshape(np.shape(a))


# --------------------------------------------------------------------------- futils restatements
def addpnt(x, y, xnew, ynew):
    """futils/TUV `addpnt`: insert (xnew, ynew) into the ascending table (x, y).  A point that is
    already present is an error, as is unsorted input."""
    x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
    if np.any(np.diff(x) < 0.0):
        raise ValueError("addpnt: x-data must be in ascending order")
    if np.any(x == xnew):
        raise ValueError("addpnt: duplicate abscissa")
    i = int(np.searchsorted(x, xnew))
    return np.insert(x, i, xnew), np.insert(y, i, ynew)


def inter2(xg, x, y):
    """futils/TUV `inter2`: map point data (x, y), connected linearly, onto the bins with edges
    `xg`: each bin receives the mean of the curve over the bin (trapezoid areas / bin width).
    The data must span the grid."""
    xg, x, y = (np.asarray(a, dtype=float) for a in (xg, x, y))
    if np.any(np.diff(xg) <= 0.0) or np.any(np.diff(x) < 0.0):
        raise ValueError("inter2: grids must be ascending")
    if x[0] > xg[0] or x[-1] < xg[-1]:
        raise ValueError("inter2: data do not span grid")
    out = np.zeros(len(xg) - 1)
    n = len(x)
    k = 0
    for i in range(len(xg) - 1):
        xgl, xgu = xg[i], xg[i + 1]
        while k < n - 1 and x[k + 1] <= xgl:
            k += 1
        area = 0.0
        j = k
        while j < n - 1 and x[j] < xgu:
            a1, a2 = max(x[j], xgl), min(x[j + 1], xgu)
            if x[j + 1] != x[j] and a2 > a1:
                slope = (y[j + 1] - y[j]) / (x[j + 1] - x[j])
                b1 = y[j] + slope * (a1 - x[j])
                b2 = y[j] + slope * (a2 - x[j])
                area += (a2 - a1) * (b2 + b1) / 2.0
            j += 1
        out[i] = area / (xgu - xgl)
    return out


def _pad_and_bin(wavl, x, y, pad_value):
    """The reference's addpnt x4 + inter2 idiom (types_create.f90:54-63, :1185-1197): hold
    `pad_value` just outside the data range and out to 0 and huge."""
    try:
        x, y = addpnt(x, y, x[0] * (1.0 - RDELTA), pad_value)
        x, y = addpnt(x, y, 0.0, pad_value)
        x, y = addpnt(x, y, x[-1] * (1.0 + RDELTA), pad_value)
        x, y = addpnt(x, y, HUGE, pad_value)
        return inter2(wavl, x, y)
    except ValueError:
        return None


def interp_discrete_to_bins(wavl, x, y, mode, fill_value=None):
    """futils `interp_discrete_to_bins(bins, x, y, out, mode, fill)`: bin means of the linearly
    connected points; beyond the data 'Constant' holds the end values, 'FillValue' holds
    `fill_value` (restated; parity unpinned)."""
    x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
    if np.any(np.diff(x) <= 0.0):
        raise ClimaException("interp_discrete_to_bins: `x` must be strictly increasing")
    if mode == "Constant":
        xx = np.concatenate([[min(0.0, x[0] - 1.0)], x, [HUGE]]) if x[0] > 0.0 else np.concatenate([x, [HUGE]])
        yy = np.concatenate([[y[0]], y, [y[-1]]]) if x[0] > 0.0 else np.concatenate([y, [y[-1]]])
        return inter2(wavl, xx, yy)
    if mode == "FillValue":
        out = _pad_and_bin(wavl, x, y, fill_value)
        if out is None:
            raise ClimaException("interp_discrete_to_bins: interpolation failed")
        return out
    raise ClimaException('interp_discrete_to_bins: unknown mode "%s"' % mode)


def rayleigh_vardavas(A, B, Delta, lam_nm):
    """src/clima_eqns.f90:240-246 (the powers through the C library's pow, element by element: see _pow10)"""
    import math
    lam = np.asarray(lam_nm, dtype=float)
    out = np.array([(4.577e-21 * ((6.0 + 3.0 * Delta) / (6.0 - 7.0 * Delta)) *
                     math.pow(A * (1.0 + B / math.pow(float(v) * 1.0e-3, 2.0)), 2.0) * (1.0 / math.pow(float(v) * 1.0e-3, 4.0)))
                    for v in lam.ravel()])
    return out.reshape(lam.shape) if lam.ndim else float(out[0])


# --------------------------------------------------------------------------- pieces
def read_stellar_flux(star_file, wavl):
    """types_create.f90:9-78: text table (one header line; wavelength nm, flux mW/m^2/nm) ->
    mW/m^2/Hz per bin of `wavl`."""
    try:
        d = np.loadtxt(star_file, skiprows=1, ndmin=2)
    except OSError:
        raise ClimaException("The input file " + star_file + " does not exist.")
    except ValueError:
        raise ClimaException("Problem reading " + star_file)
    flux = _pad_and_bin(wavl, d[:, 0], d[:, 1], 0.0)
    if flux is None:
        raise ClimaException("Problem interpolating " + star_file.strip())
    wavl = np.asarray(wavl, dtype=float)
    wavl_av = 0.5 * (wavl[:-1] + wavl[1:])
    return flux * (((wavl_av * 1.0e-9) * wavl_av) / C_LIGHT)   # :70-76


def _check_dataset(h, name, ndims, prefix):
    """check_h5_dataset, types_create.f90:1380-1405"""
    if not h.exists(name):
        raise ClimaException('%s: dataset "%s" does not exist' % (prefix, name))
    if len(h.shape(name)) != ndims:
        raise ClimaException('%s: dataset "%s" has wrong number of dimensions' % (prefix, name))
    if not h.is_float(name):
        raise ClimaException('%s: dataset "%s" has the wrong type' % (prefix, name))


def read_wavl(filename, channel):
    """types_create.f90:647-687; `channel` "ir" or "sol" -> bin edges in nm."""
    if not h5lite.is_hdf5(filename):
        raise ClimaException('Failed to read "' + filename + '".')
    name = channel + "_wavl"
    with h5lite.File(filename) as h:
        _check_dataset(h, name, 1, filename + "/" + name)
        return h.read(name) * 1.0e3


def read_ktable(filename, sp_ind):
    """create_Ktable, types_create.f90:1265-1378.  `log10k` is declared
    (ngauss, npress, ntemp, nwav) in Fortran, i.e. C shape (nwav, ntemp, npress, ngauss): exactly
    the [bin][T][P][g] order the device tables use, so the bytes pass through unchanged."""
    if not h5lite.is_hdf5(filename):
        raise ClimaException('Failed to read "' + filename + '".')
    with h5lite.File(filename) as h:
        for name, nd in (("weights", 1), ("log10P", 1), ("T", 1), ("wavelengths", 1), ("log10k", 4)):
            _check_dataset(h, name, nd, filename)
        weights, log10P, temp = h.read("weights"), h.read("log10P"), h.read("T")
        wavl = h.read("wavelengths") * 1.0e3
        log10k = h.read("log10k")
    want = (len(wavl) - 1, len(temp), len(log10P), len(weights))
    if log10k.shape != want:
        raise ClimaException('"log10k" has a bad dimension in "%s"' % filename)
    if np.any(np.diff(log10P) <= 0.0) or np.any(np.diff(temp) <= 0.0) or len(log10P) < 2 or len(temp) < 2:
        raise ClimaException('Failed to initialize interpolator for "%s". Error code:   1' % filename)
    return dict(sp_ind=sp_ind, weights=weights, log10P=log10P, temp=temp, log10k=log10k), wavl


def _regrid_rows(filename, wavl, wav_f, rows):
    """rows [nT][nwav_file] of log10 values -> [nw][nT] on the bins (types_create.f90:1216-1240)."""
    out = np.empty((len(wavl) - 1, rows.shape[0]))
    for i in range(rows.shape[0]):
        r = _pad_and_bin(wavl, wav_f, rows[i], LOG10TINY)
        if r is None:
            raise ClimaException('Problem interpolating data in "%s"' % filename.strip())
        out[:, i] = r
    return out


def read_h5_xsection(filename, wavl, xs_type, sp1, sp2=-1):
    """read_h5_Xsection, types_create.f90:1105-1263: `log10xs` 1-D (no T dependence) or
    (ntemp, nwav) in Fortran = C (nwav, ntemp)."""
    if not h5lite.is_hdf5(filename):
        raise ClimaException('Failed to read "' + filename + '".')
    with h5lite.File(filename) as h:
        if not h.exists("log10xs"):
            raise ClimaException(filename + ': dataset "log10xs" does not exist')
        dim = len(h.shape("log10xs")) - 1
        if dim not in (0, 1):
            raise ClimaException("Issue reading " + filename)
        _check_dataset(h, "wavelengths", 1, filename)
        wav_f = h.read("wavelengths") * 1.0e3
        if dim == 0:
            _check_dataset(h, "log10xs", 1, filename)
            r = _pad_and_bin(wavl, wav_f, h.read("log10xs"), LOG10TINY)
            if r is None:
                raise ClimaException('Problem interpolating data in "%s"' % filename.strip())
            return dict(xs_type=xs_type, dim=0, sp1=sp1, sp2=sp2, temp=None, data=_pow10(r))
        _check_dataset(h, "T", 1, filename)
        temp = h.read("T")
        _check_dataset(h, "log10xs", 2, filename)
        raw = h.read("log10xs")           # C (nwav_file, ntemp)
    if raw.shape[1] != len(temp):
        raise ClimaException('"log10xs" has a bad dimension in "%s"' % filename.strip())
    data = _regrid_rows(filename, wavl, wav_f, np.ascontiguousarray(raw.T))
    if len(temp) < 2 or np.any(np.diff(temp) <= 0.0):
        raise ClimaException('Failed to initialize interpolator for "%s"' % filename)
    return dict(xs_type=xs_type, dim=1, sp1=sp1, sp2=sp2, temp=temp, data=data)


def read_water_continuum(model, filename, species_names, wavl):
    """create_WaterContinuum, types_create.f90:868-1046"""
    if "H2O" not in species_names:
        raise ClimaException('"H2O" must be a species to include the "continuum" opacity')
    if not len(species_names) > 1:
        raise ClimaException('There must be more than 1 species in order to use the "continuum" opacity')
    if not h5lite.is_hdf5(filename):
        raise ClimaException('Continuum "' + model + '" is not avaliable.')
    with h5lite.File(filename) as h:
        _check_dataset(h, "wavelengths", 1, filename)
        wav_f = h.read("wavelengths") * 1.0e3
        _check_dataset(h, "T", 1, filename)
        temp = h.read("T")
        out = {}
        for name in ("log10xs_H2O", "log10xs_foreign"):
            _check_dataset(h, name, 2, filename)
            raw = h.read(name)             # C (nwav_file, ntemp)
            if raw.shape[1] != len(temp):
                raise ClimaException('"%s" has a bad dimension in "%s"' % (name, filename.strip()))
            out[name] = _regrid_rows(filename, wavl, wav_f, np.ascontiguousarray(raw.T))
    if len(temp) < 2 or np.any(np.diff(temp) <= 0.0):
        raise ClimaException('Failed to initialize interpolator for "%s"' % filename)
    return dict(LH2O=list(species_names).index("H2O"), temp=temp, log10_H2O=out["log10xs_H2O"],
                log10_foreign=out["log10xs_foreign"], model=model)


def read_photolysis_xsection(filename, sp, sp_ind, wavl):
    """create_PhotolysisXsection, types_create.f90:1407-1468 (wavelengths in nm here)."""
    if not h5lite.is_hdf5(filename):
        raise ClimaException('Species "' + sp + '" does not have photolysis xsection data')
    with h5lite.File(filename) as h:
        _check_dataset(h, "wavelengths", 1, filename)
        wv = h.read("wavelengths")
        _check_dataset(h, "photoabsorption", 1, filename)
        xs = h.read("photoabsorption")
    xs = _log10(np.maximum(xs, np.finfo(np.float64).tiny))
    return dict(xs_type=XS_PHOTOLYSIS, dim=0, sp1=sp_ind, sp2=-1, temp=None,
                data=_pow10(interp_discrete_to_bins(wavl, wv, xs, "FillValue", LOG10TINY)))


def read_particle_xsection(filename, p_ind, dat_name, wavl):
    """create_ParticleXsection, types_create.f90:734-866: radii um -> cm; w0, qext, g0 declared
    (nrad, nwav) in Fortran = C (nwav, nrad)."""
    if not h5lite.is_hdf5(filename):
        raise ClimaException("Was unable to open mie data file " + filename.strip())
    with h5lite.File(filename) as h:
        _check_dataset(h, "wavelengths", 1, filename)
        wv = h.read("wavelengths")
        _check_dataset(h, "radii", 1, filename)
        radii = h.read("radii") / 1.0e4
        raw = {}
        for name in ("w0", "qext", "g0"):
            _check_dataset(h, name, 2, filename)
            raw[name] = h.read(name)
            if raw[name].shape != (len(wv), len(radii)):
                raise ClimaException('"%s" has the wrong shape in "%s"' % (name, filename))
    out = {}
    for name in ("w0", "qext", "g0"):
        a = np.empty((len(wavl) - 1, len(radii)))
        for i in range(len(radii)):
            a[:, i] = interp_discrete_to_bins(wavl, wv, raw[name][:, i], "Constant")
        out[name] = a
    if len(radii) < 2 or np.any(np.diff(radii) <= 0.0):
        raise ClimaException('Failed to initialize interpolator for "%s"' % filename)
    return dict(p_ind=p_ind, radii=radii, w0=out["w0"], qext=out["qext"], gt=out["g0"], dat_name=dat_name)


# --------------------------------------------------------------------------- settings
class SettingsOpacity:
    """unpack_settingsopacity, src/clima_types_create.f90:799-996.  Every opacity key is either
    absent (None), a bool ("on": everything the data directory has), or an explicit list."""

    def __init__(self, op_dict, filename="settings"):
        if "opacities" not in op_dict:
            raise ClimaException(filename + ': "opacities" is required in "optical-properties"')
        o = op_dict["opacities"]
        self.k_method = None
        self.k_distributions = self.cia = self.rayleigh = self.photolysis_xs = None
        self.water_continuum = None
        self.particle_xs = None
        if "k-distributions" in o:
            self.k_method = str(op_dict.get("k-method", "")).strip()
            if self.k_method != "RandomOverlapResortRebin":
                raise ClimaException('k-method "%s" in "%s" is not an option.' % (self.k_method, filename))
            self.k_distributions = self._list_or_bool(o["k-distributions"], "k-distributions")
        for key, attr in (("CIA", "cia"), ("rayleigh", "rayleigh"), ("photolysis-xs", "photolysis_xs")):
            if key in o:
                setattr(self, attr, self._list_or_bool(o[key], key))
        if "water-continuum" in o:
            self.water_continuum = str(o["water-continuum"]).strip()
        if o.get("particle-xs") is not None:
            self.particle_xs = []
            for it in o["particle-xs"]:
                if not isinstance(it, dict):
                    raise ClimaException('"particle-xs" entries must be dictionaries.')
                self.particle_xs.append((str(it["name"]).strip(), str(it["data"]).strip()))
            names = [n for n, _ in self.particle_xs]
            for n in names:
                if names.count(n) > 1:
                    raise ClimaException('"%s" is a duplicate in particle-xs' % n)

    @staticmethod
    def _list_or_bool(node, key):
        if isinstance(node, (list, tuple)):
            lst = [str(x).strip() for x in node]
            for x in lst:
                if lst.count(x) > 1:
                    raise ClimaException('"%s" is a duplicate in %s' % (x, key))
            return lst
        if isinstance(node, bool):
            return node
        if isinstance(node, str) and node.lower() in ("on", "off", "true", "false", "yes", "no"):
            return node.lower() in ("on", "true", "yes")
        raise ClimaException('"%s" must be a list or a scalar.' % key)


def read_settings(settings_file):
    """The parts of a Clima settings YAML that `Radtran` needs (clima_types_create.f90:578-600,
    :737-797): optical-properties/{species/{gases,particles}, opacities, k-method,
    wavelength-bins-file}."""
    import yaml
    try:
        with open(settings_file) as f:
            root = yaml.safe_load(f)
    except OSError:
        raise ClimaException('Could not open "%s"' % settings_file)
    if not isinstance(root, dict) or "optical-properties" not in root:
        raise ClimaException('%s: "optical-properties" is required' % settings_file)
    op = root["optical-properties"]
    sp = op.get("species") or {}
    gases = [str(s) for s in (sp.get("gases") or [])]
    particles = [str(s) for s in (sp.get("particles") or [])]
    return dict(gases=gases, particles=particles, sop=SettingsOpacity(op, settings_file),
                wavelength_bins_file=op.get("wavelength-bins-file"), root=root)


def parse_cia_pair(pair, species_names):
    """parse_cia_pair, types_create.f90:689-732: split at the '-' that leaves two known species."""
    pair = pair.strip()
    if len(pair) < 2:
        raise ClimaException('Could not parse CIA species pair "%s"' % pair)
    matches = []
    for p in range(1, len(pair) - 1):
        if pair[p] != "-":
            continue
        left, right = pair[:p].strip(), pair[p + 1:].strip()
        if left and right and left in species_names and right in species_names:
            matches.append((species_names.index(left), species_names.index(right)))
    if not matches:
        raise ClimaException('Could not parse CIA species pair "%s" into two known species.' % pair)
    if len(matches) > 1:
        raise ClimaException('CIA species pair "%s" is ambiguous; matched multiple species splits.' % pair)
    return matches[0]


# --------------------------------------------------------------------------- the whole table set
class LoadedTables:
    """Same shape as clima_amd.synthetic.TableSet (what `Radtran(tables, ...)` consumes)."""

    def __init__(self):
        self.species_names, self.particle_names = (), ()
        self.wavl = None
        self.ktables, self.xsections, self.particles = [], [], []
        self.continuum = None
        self.ir_wavl = self.sol_wavl = self.photons_sol = None
        self.k_method_name = "RandomOverlapResortRebin"

    nw = property(lambda s: len(s.wavl) - 1)
    nsp = property(lambda s: len(s.species_names))
    np_ = property(lambda s: len(s.particle_names))
    ng = property(lambda s: len(s.ktables[0]["weights"]))


def _is_close(a, b, tol):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.abs(a - b) <= tol * np.maximum(np.abs(a), np.abs(b))


def create_optical_properties(datadir, species_names, particle_names, sop):
    """create_OpticalProperties, types_create.f90:272-645"""
    species_names, particle_names = list(species_names), list(particle_names)
    t = LoadedTables()
    t.species_names, t.particle_names = tuple(species_names), tuple(particle_names)
    J = os.path.join

    # ---- k-distributions (:298-390)
    kd = sop.k_distributions
    if kd is None or kd is False:
        raise ClimaException("You must specify at least one k-distribution in the settings file.")
    if kd is True:
        kd = [s for s in species_names if os.path.exists(J(datadir, "kdistributions", s + ".h5"))]
        if not kd:
            raise ClimaException("No k-distribution data was found, but at least one k-distribution is needed.")
    for i, sp in enumerate(kd):
        if sp not in species_names:
            raise ClimaException('Species "%s" in optical property "k-distributions" is not in the list of species.' % sp)
        k, wavl = read_ktable(J(datadir, "kdistributions", sp + ".h5"), species_names.index(sp))
        if i == 0:
            t.wavl = wavl
        elif len(wavl) != len(t.wavl) or not np.all(_is_close(t.wavl, wavl, 1.0e-7)):
            raise ClimaException('Species "%s" has wavelength bins that do not match the wavelength bins '
                                 'for other species' % sp)
        t.ktables.append(k)
    for k in t.ktables[1:]:
        if len(k["weights"]) != len(t.ktables[0]["weights"]) or \
                not np.all(_is_close(t.ktables[0]["weights"], k["weights"], 1.0e-12)):
            raise ClimaException("All k-coeff bin weights must match.")
    t.k_method_name = sop.k_method

    # ---- CIA (:395-471)
    cia_list = []
    if sop.cia is not None and sop.cia is not False:
        if sop.cia is True:
            for a in species_names:
                for b in species_names:
                    if os.path.exists(J(datadir, "CIA", a + "-" + b + ".h5")) and \
                            not (sop.water_continuum is not None and "H2O" in (a, b)):
                        cia_list.append(a + "-" + b)
        else:
            cia_list = list(sop.cia)
        for pair in cia_list:
            i1, i2 = parse_cia_pair(pair, species_names)
            t.xsections.append(read_h5_xsection(J(datadir, "CIA", pair + ".h5"), t.wavl, XS_CIA, i1, i2))

    # ---- Rayleigh (:476-541)
    if sop.rayleigh is not None and sop.rayleigh is not False:
        import yaml
        fn = J(datadir, "rayleigh", "rayleigh.yaml")
        try:
            with open(fn) as f:
                root = yaml.safe_load(f)
        except OSError:
            raise ClimaException('Could not open "%s"' % fn)
        if not isinstance(root, dict):
            raise ClimaException('There is an issue with formatting in "%s"' % fn)
        names = [k for k in root if k in species_names] if sop.rayleigh is True else list(sop.rayleigh)
        lam = t.wavl[:-1]   # the reference evaluates at the lower bin edge (:1083-1085)
        for sp in names:
            if sp not in species_names:
                raise ClimaException('Species "%s" in optical property "rayleigh" is not in the list of species.' % sp)
            try:
                d = root[sp]["data"]
                A, B, Delta = float(d["A"]), float(d["B"]), float(d["Delta"])
            except (KeyError, TypeError, ValueError):
                raise ClimaException('%s: Rayleigh data for "%s" is missing or malformed' % (fn, sp))
            t.xsections.append(dict(xs_type=XS_RAYLEIGH, dim=0, sp1=species_names.index(sp), sp2=-1, temp=None,
                                    data=rayleigh_vardavas(A, B, Delta, lam)))

    # ---- photolysis cross sections (:546-591)
    if sop.photolysis_xs is not None and sop.photolysis_xs is not False:
        names = [s for s in species_names if os.path.exists(J(datadir, "xsections", s + ".h5"))] \
            if sop.photolysis_xs is True else list(sop.photolysis_xs)
        for sp in names:
            if sp not in species_names:
                raise ClimaException('Species "%s" in optical property "photolysis-xs" is not in the list of species.' % sp)
            t.xsections.append(read_photolysis_xsection(J(datadir, "xsections", sp + ".h5"), sp,
                                                        species_names.index(sp), t.wavl))

    # ---- particles (:596-614)
    for name, dat in (sop.particle_xs or []):
        if name not in particle_names:
            raise ClimaException('Species "%s" in optical property "particle-xs" is not in the list of particles.' % name)
        t.particles.append(read_particle_xsection(J(datadir, "aerosol_xsections", dat, "mie_" + dat + ".h5"),
                                                  particle_names.index(name), dat, t.wavl))

    # ---- water continuum (:619-642)
    if sop.water_continuum is not None:
        for pair in cia_list:
            j = pair.find("-")
            if "H2O" in (pair[:j], pair[j + 1:]):
                raise ClimaException('Optical property "water-continuum" is set, but CIA "%s" is also set. This is '
                                     'not allowed because it would double count opacity.' % pair)
        t.continuum = read_water_continuum(sop.water_continuum, J(datadir, "water_continuum", sop.water_continuum + ".h5"),
                                           species_names, t.wavl)
    return t


def create_rt_channel(datadir, channel, wavelength_bins_file, wavl):
    """create_RTChannel, types_create.f90:226-270 -> the channel's bin edges (a contiguous slice
    of the opacity grid)."""
    fn = wavelength_bins_file if wavelength_bins_file else os.path.join(datadir, "kdistributions", "bins.h5")
    w = read_wavl(fn, channel)
    i1 = int(np.argmin(np.abs(w[0] - wavl)))
    i2 = int(np.argmin(np.abs(w[-1] - wavl)))
    if len(w) != len(wavl[i1:i2 + 1]) or not np.all(_is_close(w, wavl[i1:i2 + 1], 1.0e-7)):
        raise ClimaException('The wavelength bins "%s" are not compatible with the k-distribution wavelength bins.'
                             % fn.strip())
    return wavl[i1:i2 + 1].copy()


def load_tables(settings_file, star_file, datadir, species_names=None, particle_names=None):
    """Everything `create_Radtran_1/2` loads (clima_radtran.f90:98-219): -> LoadedTables."""
    s = read_settings(settings_file)
    species = list(species_names) if species_names is not None else s["gases"]
    particles = list(particle_names) if particle_names is not None else s["particles"]
    if not species:
        raise ClimaException('"%s/optical-properties/species" does not contain any gases' % settings_file)
    t = create_optical_properties(datadir, species, particles, s["sop"])
    t.ir_wavl = create_rt_channel(datadir, "ir", s["wavelength_bins_file"], t.wavl)
    t.sol_wavl = create_rt_channel(datadir, "sol", s["wavelength_bins_file"], t.wavl)
    t.photons_sol = read_stellar_flux(star_file, t.sol_wavl)
    return t


def radtran_from_files(settings_file, star_file, num_zenith_angles, surface_albedo, nz, datadir):
    """`Radtran(settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir, err)`
    (src/radtran/clima_radtran.f90:98-126) on the MI355X."""
    from .radtran import Radtran
    return Radtran(load_tables(settings_file, star_file, datadir), nz, num_zenith_angles, surface_albedo)
