"""Python mirror of Clima's `Radtran` (clima/cython/Radtran.pyx, ClimaRadtranWrk.pyx,
RTChannel.pyx) over the HIP C ABI.  Same attribute names, argument meaning and error
behaviour (`ClimaException` carrying the reference's message text).

The reference's Python layer reaches `Radtran%radiate` only through `AdiabatClimate`
(`c.rad`); here the `Radtran` object itself is constructible from a table set because the
HDF5/YAML loaders are outside the hot path (SURVEY.md 8(f) "next #1").
"""
import ctypes as C
import os

import numpy as np

from . import lib as _lib

_dp = C.POINTER(C.c_double)


class ClimaException(Exception):
    """clima/cython/_clima.pyx: raised with the Fortran `err` text."""


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(v):
    return C.byref(C.c_int(int(v)))


def _f(v):
    return C.byref(C.c_double(float(v)))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _fo(a):
    return np.asfortranarray(a, dtype=np.float64)


class RTChannel:
    """clima/cython/RTChannel.pyx"""

    def __init__(self, L, ptr):
        self._L, self._ptr = L, ptr

    def _vec(self, name):
        n = C.c_int()
        getattr(self._L, "rtchannel_%s_get_size" % name)(self._ptr, C.byref(n))
        arr = np.empty(n.value, np.double)
        getattr(self._L, "rtchannel_%s_get" % name)(self._ptr, C.byref(n), _d(arr))
        return arr

    @property
    def wavl(self):
        "ndarray[double,ndim=1]. Edges of the wavelength bins (nm)."
        return self._vec("wavl")

    @property
    def freq(self):
        "ndarray[double,ndim=1]. Edges of the bins in frequency (1/s)."
        return self._vec("freq")


class ClimaRadtranWrk:
    """clima/cython/ClimaRadtranWrk.pyx: arrays are copied out, Fortran order, index 0 =
    ground level, index nz = top of the atmosphere."""

    def __init__(self, L, ptr):
        self._L, self._ptr = L, ptr

    def _mat(self, name):
        d1, d2 = C.c_int(), C.c_int()
        getattr(self._L, "climaradtranwrk_%s_get_size" % name)(self._ptr, C.byref(d1), C.byref(d2))
        arr = np.empty((d1.value, d2.value), np.double, order="F")
        getattr(self._L, "climaradtranwrk_%s_get" % name)(self._ptr, C.byref(d1), C.byref(d2), _d(arr))
        return arr

    def _vec(self, name):
        d1 = C.c_int()
        getattr(self._L, "climaradtranwrk_%s_get_size" % name)(self._ptr, C.byref(d1))
        arr = np.empty(d1.value, np.double)
        getattr(self._L, "climaradtranwrk_%s_get" % name)(self._ptr, C.byref(d1), _d(arr))
        return arr

    fup_a = property(lambda self: self._mat("fup_a"), doc="(nz+1,nw) mW/m^2/Hz upward flux per bin")
    fdn_a = property(lambda self: self._mat("fdn_a"), doc="(nz+1,nw) mW/m^2/Hz downward flux per bin")
    amean = property(lambda self: self._mat("amean"), doc="(nz+1,nw) mean intensity, photons/cm^2/s (solar)")
    tau_band = property(lambda self: self._mat("tau_band"), doc="(nz,nw) band optical thickness")
    fup_n = property(lambda self: self._vec("fup_n"), doc="(nz+1) mW/m^2 upward flux")
    fdn_n = property(lambda self: self._vec("fdn_n"), doc="(nz+1) mW/m^2 downward flux")


class Radtran:
    """`type Radtran` (src/radtran/clima_radtran.f90:31-85) on the MI355X."""

    def __init__(self, tables, nz, num_zenith_angles, surface_albedo):
        """Equivalent of `Radtran(species, particles, settings, star, num_zenith_angles,
        surface_albedo, nz, datadir)` (clima_radtran.f90:128-219) with the loaded tables
        handed over as a `clima_amd.synthetic.TableSet`-shaped object."""
        L = _lib.load()
        self._L = L
        self._ptr = C.c_void_p()
        L.allocate_radtran(C.byref(self._ptr))
        self._err = C.create_string_buffer(_lib.ERR_LEN + 1)
        t = tables
        self.nz = int(nz)
        self.ng = t.nsp  # number of gases, as Radtran%ng
        self.np = t.np_
        self.species_names = list(t.species_names)
        self.particle_names = list(t.particle_names)
        wavl = _c(t.wavl)
        L.radtran_create_begin(self._ptr, _i(nz), _i(t.nsp), _i(t.np_), _i(t.nw), _d(wavl), self._err)
        self._check()
        for k in t.ktables:
            w, lp, tt, kk = _c(k["weights"]), _c(k["log10P"]), _c(k["temp"]), _c(k["log10k"])
            L.radtran_add_ktable(self._ptr, _i(k["sp_ind"] + 1), _i(len(w)), _d(w), _i(len(lp)), _d(lp),
                                 _i(len(tt)), _d(tt), _d(kk), self._err)
            self._check()
        for x in t.xsections:
            temp = x.get("temp")
            tt = _c(temp if temp is not None else np.zeros(1))
            da = _c(x["data"])
            L.radtran_add_xsection(self._ptr, _i(x["xs_type"]), _i(x["dim"]), _i(x["sp1"] + 1),
                                   _i(x.get("sp2", -1) + 1), _i(0 if temp is None else len(temp)), _d(tt), _d(da),
                                   self._err)
            self._check()
        if t.continuum is not None:
            c = t.continuum
            tt, a, b = _c(c["temp"]), _c(c["log10_H2O"]), _c(c["log10_foreign"])
            L.radtran_set_water_continuum(self._ptr, _i(c["LH2O"] + 1), _i(len(tt)), _d(tt), _d(a), _d(b), self._err)
            self._check()
        for p in t.particles:
            r, a, b, g = _c(p["radii"]), _c(p["w0"]), _c(p["qext"]), _c(p["gt"])
            L.radtran_add_particle(self._ptr, _i(p["p_ind"] + 1), _i(len(r)), _d(r), _d(a), _d(b), _d(g), self._err)
            self._check()
        iw, sw = _c(t.ir_wavl), _c(t.sol_wavl)
        L.radtran_set_channels(self._ptr, _i(len(iw)), _d(iw), _i(len(sw)), _d(sw), self._err)
        self._check()
        ps = _c(t.photons_sol)
        L.radtran_set_photons_sol(self._ptr, _i(len(ps)), _d(ps), self._err)
        self._check()
        L.radtran_create_end(self._ptr, _i(num_zenith_angles), _f(surface_albedo), self._err)
        self._check()
        self.nw = t.nw
        self.ngauss = t.ng
        # names for opacities2yaml (clima_radtran_types.f90:328-430)
        L.radtran_set_names(self._ptr, "\n".join(self.species_names).encode(), "\n".join(self.particle_names).encode(),
                            self._err)
        self._check()
        cont = t.continuum.get("model", "MT_CKD") if t.continuum is not None else ""
        L.radtran_set_opacity_labels(self._ptr, str(getattr(t, "k_method_name", "RandomOverlapResortRebin")).encode(),
                                     str(cont).encode(),
                                     "\n".join(str(p.get("dat_name", "unknown")) for p in t.particles).encode(), self._err)
        self._check()

    @classmethod
    def from_files(cls, settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir):
        """`Radtran(settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir, err)`
        (src/radtran/clima_radtran.f90:98-126): species and opacities from the settings YAML, tables
        from a `photochem_clima_data`-style directory (clima_amd/data_loader.py)."""
        from . import data_loader
        return cls(data_loader.load_tables(settings_f, star_f, datadir), nz, num_zenith_angles, surface_albedo)

    @classmethod
    def from_files_c(cls, settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir):
        """The same constructor with the files read BEHIND the C ABI (radtran_create_from_files,
        clima_amd/csrc/radtran_loader.hip): what a Fortran / C host without the reference's loaders calls."""
        self = cls.__new__(cls)
        L = _lib.load()
        self._L = L
        self._ptr = C.c_void_p()
        L.allocate_radtran(C.byref(self._ptr))
        self._err = C.create_string_buffer(_lib.ERR_LEN + 1)
        L.radtran_create_from_files(self._ptr, str(settings_f).encode(), str(star_f).encode(), _i(num_zenith_angles),
                                    _f(surface_albedo), _i(nz), str(datadir).encode(), self._err)
        self._check()
        d = [C.c_int() for _ in range(5)]
        L.radtran_dims_get(self._ptr, *[C.byref(x) for x in d])
        self.nz, self.ng, self.np, self.nw, self.ngauss = (x.value for x in d)
        a, b = C.create_string_buffer(4096), C.create_string_buffer(4096)
        L.radtran_names_get(self._ptr, _i(4096), a, b)
        self.species_names = [x for x in a.value.decode().split("\n") if x]
        self.particle_names = [x for x in b.value.decode().split("\n") if x]
        return self

    def __del__(self):
        if getattr(self, "_ptr", None) is not None and self._ptr.value:
            if getattr(self, "_locked", None):
                self._L.radtran_spectra_release(self._ptr)
            self._L.deallocate_radtran(self._ptr)
            self._ptr = C.c_void_p()

    def _check(self):
        msg = self._err.value
        if len(msg.strip()) > 0:
            raise ClimaException(msg.decode("utf-8").strip())

    # ------------------------------------------------------------------ the path
    def _column_args(self, T, P, densities, dz, pdensities, radii):
        T, P, dz = _c(T), _c(P), _c(dz)
        densities = _fo(densities)
        if densities.ndim != 2:
            raise ClimaException('"densities" has the wrong input dimension.')
        has_p = pdensities is not None or radii is not None
        if has_p and (pdensities is None or radii is None):
            raise ClimaException("Both pdensities and radii must be arguments.")
        if has_p:
            pdensities, radii = _fo(pdensities), _fo(radii)
            p1, p2 = pdensities.shape if pdensities.ndim == 2 else (pdensities.shape[0], 1)
            r1, r2 = radii.shape if radii.ndim == 2 else (radii.shape[0], 1)
            pd, ra = _d(pdensities), _d(radii)
        else:
            p1 = p2 = r1 = r2 = 0
            pd = ra = None
        keep = (T, P, dz, densities, pdensities, radii)
        args = (_i(len(T)), _d(T), _i(len(P)), _d(P), _i(densities.shape[0]), _i(densities.shape[1]), _d(densities),
                _i(len(dz)), _d(dz), _i(1 if has_p else 0), _i(p1), _i(p2), pd, _i(r1), _i(r2), ra)
        return keep, args

    def radiate(self, T_surface, T, P, densities, dz, pdensities=None, radii=None, compute_solar=True,
                compute_opacity=True):
        """Radtran%radiate (clima_radtran.f90:221-318).  T (K), P (bar), densities (nz,ng)
        molecules/cm^3, dz (cm); fills wrk_ir, wrk_sol and f_total."""
        keep, a = self._column_args(T, P, densities, dz, pdensities, radii)
        self._L.radtran_radiate_wrapper(self._ptr, _f(T_surface), *a, _i(compute_solar), _i(compute_opacity),
                                        self._err)
        del keep
        self._check()

    def TOA_fluxes(self, T_surface, T, P, densities, dz, pdensities=None, radii=None, compute_solar=True,
                   compute_opacity=True):
        """Radtran%TOA_fluxes (clima_radtran.f90:320-342) -> (ISR, OLR) in mW/m^2."""
        keep, a = self._column_args(T, P, densities, dz, pdensities, radii)
        isr, olr = C.c_double(), C.c_double()
        self._L.radtran_toa_fluxes_wrapper(self._ptr, _f(T_surface), *a, _i(compute_solar), _i(compute_opacity),
                                           C.byref(isr), C.byref(olr), self._err)
        del keep
        self._check()
        return isr.value, olr.value

    def bench_toa_fluxes(self, n, T_surface, T, P, densities, dz, pdensities=None, radii=None):
        """`n` synchronous TOA_fluxes calls timed one by one INSIDE the library (us per call, numpy array):
        the drop-in call as a Fortran / C host sees it, without the ctypes layer's own cost."""
        keep, a = self._column_args(T, P, densities, dz, pdensities, radii)
        us = np.zeros(int(n))
        isr, olr = C.c_double(), C.c_double()
        self._L.clima_bench_toa_fluxes(self._ptr, _i(n), _f(T_surface), *a, _d(us), C.byref(isr), C.byref(olr), self._err)
        del keep
        self._check()
        return us

    def bench_resident_sync(self, n):
        """`n` times radiate_resident + synchronize, timed one by one inside the library (us per call)."""
        us = np.zeros(int(n))
        self._L.clima_bench_resident_sync(self._ptr, _i(n), _d(us), self._err)
        self._check()
        return us

    def bench_resident_graph(self, n, k=1):
        """Timing only: one resident call's launches as a hipGraph, `n` passes of `k` graph launches + synchronize (us
        per pass); the replayed results are not valid."""
        us = np.zeros(int(n))
        self._L.clima_bench_resident_graph(self._ptr, _i(n), _i(k), _d(us), self._err)
        self._check()
        return us

    def apply_radiation_enhancement(self, rad_enhancement):
        self._L.radtran_apply_radiation_enhancement(self._ptr, _f(rad_enhancement))

    def opacities2yaml(self):
        """Radtran%opacities2yaml (clima/cython/Radtran.pyx:68-81): YAML text naming every opacity."""
        n, cp = C.c_int(), C.c_void_p()
        self._L.radtran_opacities2yaml_wrapper_1(self._ptr, C.byref(n), C.byref(cp))
        buf = C.create_string_buffer(n.value + 1)
        self._L.radtran_opacities2yaml_wrapper_2(self._ptr, C.byref(cp), C.byref(n), buf)
        return buf.value.decode()

    def set_custom_optical_properties(self, wv, P, dtau_dz, w0, g0):
        """Radtran%set_custom_optical_properties (src/radtran/clima_radtran.f90:494-506): `wv` nm,
        `P` dynes/cm^2 (decreasing), `dtau_dz` 1/cm, `w0`, `g0` of shape (size(P), size(wv))."""
        wv, P = _c(wv), _c(P)
        arrs = [np.asfortranarray(x, dtype=np.float64) for x in (dtau_dz, w0, g0)]
        for a in arrs:
            if a.ndim != 2:
                raise ClimaException("`dtau_dz`, `w0` and `g0` must be 2-D arrays")
        args = []
        for a in arrs:
            args += [_i(a.shape[0]), _i(a.shape[1]), _d(a)]
        self._L.radtran_set_custom_optical_properties(self._ptr, _i(len(wv)), _d(wv), _i(len(P)), _d(P), *args, self._err)
        self._check()

    def unset_custom_optical_properties(self):
        """src/radtran/clima_radtran.f90:508-512"""
        self._L.radtran_unset_custom_optical_properties(self._ptr)

    def TOA_fluxes_batch(self, columns, return_fluxes=False):
        """Many independent columns (BASELINE config 4): `columns` is a sequence of column dicts /
        `synthetic.Column`s (T_surface, T, P, densities, dz[, pdensities, radii]).  Returns
        (ISR, OLR) arrays, plus the level fluxes (nz+1, 5, ncol) = ir up, ir down, solar up,
        solar down, f_total when `return_fluxes`."""
        n = len(columns)
        nz = self.nz
        Ts = np.array([float(c["T_surface"]) for c in columns])
        T = np.stack([np.asarray(c["T"], dtype=np.float64) for c in columns])           # (ncol, nz)
        P = np.stack([np.asarray(c["P"], dtype=np.float64) for c in columns])
        dz = np.stack([np.asarray(c["dz"], dtype=np.float64) for c in columns])
        dens = np.stack([np.asarray(c["densities"], dtype=np.float64).T for c in columns])  # (ncol, nsp, nz)
        if T.shape != (n, nz) or dens.shape != (n, self.ng, nz):
            raise ClimaException('"densities" has the wrong input dimension.')
        hp = self.np > 0
        if hp:
            if any(c.get("radii") is None for c in columns):
                raise ClimaException('"pdensities" and "radii" are required arguments.')
            pd = np.stack([np.asarray(c["pdensities"], dtype=np.float64).T for c in columns])
            ra = np.stack([np.asarray(c["radii"], dtype=np.float64).T for c in columns])
        else:
            pd = ra = np.zeros(1)
        isr, olr = np.empty(n), np.empty(n)
        fl = np.empty((n, 5, nz + 1)) if return_fluxes else None
        self._L.radtran_toa_fluxes_batch(self._ptr, _i(n), _d(Ts), _d(np.ascontiguousarray(T)), _d(np.ascontiguousarray(P)),
                                         _d(np.ascontiguousarray(dens)), _d(np.ascontiguousarray(dz)), _i(1 if hp else 0),
                                         _d(np.ascontiguousarray(pd)), _d(np.ascontiguousarray(ra)), _d(isr), _d(olr),
                                         _d(fl) if return_fluxes else None, self._err)
        self._check()
        if return_fluxes:
            return isr, olr, np.transpose(fl, (2, 1, 0))
        return isr, olr

    def radiate_ir_batch(self, T_surface, T, out=None, pin=False):
        """ncol IR-only calls with the resident opacities in one go: column c is
        `radiate(T_surface[c], T[:, c], ..., compute_solar=False, compute_opacity=False)`
        (the RCE Jacobian's loop, src/adiabat/clima_adiabat_solve.f90:798-812).
        Returns (fup_n, fdn_n, f_total), each (nz+1, ncol).  `out` + `pin`: the caller's three arrays are page-locked from
        the second call with the same ones on and filled by the device directly (radtran_batch_pin_results_set); this
        object keeps them alive until spectra_release()."""
        T = np.asfortranarray(T, dtype=np.float64)
        Ts = _c(np.atleast_1d(T_surface))
        if T.ndim != 2 or T.shape[0] != self.nz or T.shape[1] != len(Ts):
            raise ClimaException('"T" has the wrong input dimension.')
        n = T.shape[1]
        pin = bool(pin) and out is not None
        if out is None:     # (out: three (nz+1, ncol) Fortran-ordered arrays to fill, as a caller that keeps its buffers would)
            out = [np.empty((self.nz + 1, n), order="F") for _ in range(3)]
        self._L.radtran_batch_pin_results_set(self._ptr, _i(1 if pin else 0))
        self._locked = getattr(self, "_locked", [])
        if pin and not any(out is o for o in self._locked):
            self._locked.append(out)
        self._L.radtran_radiate_ir_batch(self._ptr, _i(n), _d(Ts), _i(T.shape[0]), _i(n), _d(T),
                                         _d(out[0]), _d(out[1]), _d(out[2]), self._err)
        self._check()
        return tuple(out)

    # ---- HBM-resident form (bench / batched callers)
    def upload_column(self, T_surface, T, P, densities, dz, pdensities=None, radii=None):
        T, P, dz, densities = _c(T), _c(P), _c(dz), _fo(densities)
        if len(T) != self.nz or len(P) != self.nz or len(dz) != self.nz or densities.shape != (self.nz, self.ng):
            raise ClimaException('"densities" has the wrong input dimension.')
        pd = ra = None
        if pdensities is not None:
            pdensities, radii = _fo(pdensities), _fo(radii)
            pd, ra = _d(pdensities), _d(radii)
        self._L.radtran_upload_column(self._ptr, _f(T_surface), _d(T), _d(P), _d(densities), _d(dz), pd, ra, self._err)
        self._check()

    def radiate_resident(self, compute_solar=True, compute_opacity=True):
        self._L.radtran_radiate_resident(self._ptr, _i(compute_solar), _i(compute_opacity), self._err)
        self._check()

    def synchronize(self):
        self._L.radtran_synchronize(self._ptr, self._err)
        self._check()

    def set_bin_shard(self, rank, world):
        self._L.radtran_set_bin_shard(self._ptr, _i(rank), _i(world), self._err)
        self._check()

    # ---- the library's own multi-GPU step (include/clima_radtran_hip.h, radtran_comm_*): one process per GPU;
    # after comm_init_* every radiate / TOA_fluxes / radiate_resident of this handle works on the rank's bins and
    # ends with one RCCL all-reduce of the level fluxes on the handle's stream
    COMM_ID_BYTES = 128

    @staticmethod
    def set_device(device):
        L = _lib.load()
        err = C.create_string_buffer(_lib.ERR_LEN + 1)
        L.radtran_set_device(_i(device), err)
        if err.value:
            raise ClimaException(err.value.decode())

    @staticmethod
    def comm_unique_id():
        """rank 0: a fresh communicator id (bytes) to hand to every rank."""
        L = _lib.load()
        err = C.create_string_buffer(_lib.ERR_LEN + 1)
        buf = C.create_string_buffer(Radtran.COMM_ID_BYTES)
        L.radtran_comm_unique_id(buf, err)
        if err.value:
            raise ClimaException(err.value.decode())
        return buf.raw

    def comm_init_rank(self, nranks, rank, comm_id):
        assert len(comm_id) == self.COMM_ID_BYTES
        self._L.radtran_comm_init_rank(self._ptr, _i(nranks), _i(rank), C.create_string_buffer(bytes(comm_id), self.COMM_ID_BYTES), self._err)
        self._check()

    def comm_init_file(self, nranks, rank, path):
        self._L.radtran_comm_init_file(self._ptr, _i(nranks), _i(rank), os.fsencode(path), self._err)
        self._check()

    def comm(self):
        """(nranks, rank, all-reduces enqueued so far); nranks 0 without a communicator"""
        n, r, k = C.c_int(0), C.c_int(0), C.c_int(0)
        self._L.radtran_comm_get(self._ptr, C.byref(n), C.byref(r), C.byref(k))
        return n.value, r.value, k.value

    def comm_destroy(self):
        self._L.radtran_comm_destroy(self._ptr)

    def bin_shard(self):
        v = [C.c_int() for _ in range(6)]
        self._L.radtran_bin_shard_get(self._ptr, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    @property
    def fused(self):
        """True: opacity + two-stream of a compute_opacity call run as one grid (k_fused)."""
        v = C.c_int()
        self._L.radtran_fused_get(self._ptr, C.byref(v))
        return bool(v.value)

    @fused.setter
    def fused(self, on):
        self._L.radtran_fused_set(self._ptr, _i(1 if on else 0))

    @property
    def coop_items(self):
        """ng = 8: calls with at most this many (bin, source layer) items use the group-of-lanes opacity kernel."""
        v = C.c_int()
        self._L.radtran_coop_items_get(self._ptr, C.byref(v))
        return v.value

    @coop_items.setter
    def coop_items(self, n):
        self._L.radtran_coop_items_set(self._ptr, _i(int(n)))

    def spectra_all(self, do_solar=True, out=None):
        """The seven per-bin arrays of the last call through radtran_spectra_get_all (one synchronise; the arrays of
        `out` -- a dict from an earlier call -- are page-locked by the library on first use).  Column-major like the
        reference's holders: fup_a, fdn_a (nz+1, nw), tau_band (nz, nw) per channel, amean for the solar one."""
        nl, ni, ns = self.nz + 1, len(self.ir.freq) - 1, len(self.sol.freq) - 1
        if out is None:
            out = {"ir_fup_a": np.zeros((nl, ni), order="F"), "ir_fdn_a": np.zeros((nl, ni), order="F"),
                   "ir_tau_band": np.zeros((nl - 1, ni), order="F"), "sol_fup_a": np.zeros((nl, ns), order="F"),
                   "sol_fdn_a": np.zeros((nl, ns), order="F"), "sol_amean": np.zeros((nl, ns), order="F"),
                   "sol_tau_band": np.zeros((nl - 1, ns), order="F")}
        err = C.create_string_buffer(1025)
        names = ("ir_fup_a", "ir_fdn_a", "ir_tau_band", "sol_fup_a", "sol_fdn_a", "sol_amean", "sol_tau_band")
        self._L.radtran_spectra_get_all(self._ptr, C.byref(C.c_bool(bool(do_solar))), _i(nl), _i(ni), _i(ns),
                                        *[out[k].ctypes.data_as(C.POINTER(C.c_double)) for k in names], err)
        if err.value:
            raise ClimaException(err.value.decode())
        self._locked = getattr(self, "_locked", [])
        if not any(o is out for o in self._locked):
            self._locked.append(out)          # page-locked by the library: kept alive until spectra_release / the handle goes
        return out

    def spectra_release(self):
        """Un-page-lock what spectra_all locked (before those arrays are freed)."""
        self._L.radtran_spectra_release(self._ptr)
        self._locked = []

    @property
    def fused_spins(self):
        """Polls a two-stream block of the fused grid spends waiting for its opacity tiles (0: every wait expires and
        the call is repeated through separate launches)."""
        v = C.c_int()
        self._L.radtran_fused_spins_get(self._ptr, C.byref(v))
        return v.value

    @fused_spins.setter
    def fused_spins(self, n):
        self._L.radtran_fused_spins_set(self._ptr, _i(int(n)))

    @property
    def ir_green(self):
        """radiate_ir_batch's response form: 0 never, 1 (default) when enough columns are sparse deviations of one
        profile, 2 whenever any is."""
        m, b = C.c_int(), C.c_int()
        self._L.radtran_ir_green_get(self._ptr, C.byref(m), C.byref(b))
        return m.value

    @ir_green.setter
    def ir_green(self, mode):
        self._L.radtran_ir_green_set(self._ptr, _i(int(mode)))

    @property
    def ir_green_batches(self):
        """Batches that took the response form."""
        m, b = C.c_int(), C.c_int()
        self._L.radtran_ir_green_get(self._ptr, C.byref(m), C.byref(b))
        return b.value

    @property
    def fused_fallbacks(self):
        """Calls re-issued through the separate launches after a fused hand-off wait expired."""
        v = C.c_int()
        self._L.radtran_fused_fallbacks_get(self._ptr, C.byref(v))
        return v.value

    def flux_tensor(self):
        """The packed level fluxes [ir_up, ir_dn, sol_up, sol_dn][nz+1] as a torch CUDA tensor
        aliasing the library's buffer (the RCCL all-reduce payload of a bin-sharded run)."""
        import torch
        ptr, n = self.flux_device_ptr()

        class _Alias:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        return torch.as_tensor(_Alias(), device="cuda")

    def finish_reduced(self):
        self._L.radtran_finish_reduced(self._ptr, self._err)
        self._check()

    def flux_device_ptr(self):
        p, n = C.c_void_p(), C.c_int()
        self._L.radtran_flux_device_ptr(self._ptr, C.byref(p), C.byref(n))
        return p.value, n.value

    def stream(self):
        p = C.c_void_p()
        self._L.radtran_stream_get(self._ptr, C.byref(p))
        return p.value

    def profile(self, enable=True):
        """HIP events on the library stream: True/1 around every kernel, 2 around the dominant
        kernel (k_opacity) only (two event records per call instead of eight), False/0 off."""
        self._L.radtran_profile_set(self._ptr, _i(2 if enable == 2 and enable is not True else (1 if enable else 0)))

    def profile_stride(self, stride):
        """Events on every `stride`-th call only."""
        self._L.radtran_profile_stride_set(self._ptr, _i(int(stride)))

    def profile_reset(self):
        self._L.radtran_profile_reset(self._ptr)

    def kernel_time(self, kernel_id):
        ms, n = C.c_double(), C.c_int()
        self._L.radtran_kernel_time_get(self._ptr, _i(kernel_id), C.byref(ms), C.byref(n), self._err)
        self._check()
        return ms.value, n.value

    def algorithmic_bytes(self):
        a, b, c, d = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        self._L.radtran_algorithmic_bytes(self._ptr, C.byref(a), C.byref(b), C.byref(c), C.byref(d), self._err)
        self._check()
        return dict(tables_distinct=a.value, input=b.value, output=c.value, tables_full=d.value)

    def algorithmic_nodes(self):
        """distinct interpolation nodes the uploaded column touches (N_PT, N_T) and the table sizes."""
        v = [C.c_double() for _ in range(4)]
        self._L.radtran_algorithmic_nodes(self._ptr, *[C.byref(x) for x in v], self._err)
        self._check()
        return dict(N_PT=v[0].value, N_PT_full=v[1].value, N_T=v[2].value, N_T_full=v[3].value)

    def opr(self):
        """OpticalPropertiesResult: tau,w0 (nz,ngauss,nw), g,tau_band (nz,nw); TOA-first."""
        nz, ng, nw = self.nz, self.ngauss, self.nw
        tau = np.empty((nz, ng, nw), order="F")
        w0 = np.empty((nz, ng, nw), order="F")
        g = np.empty((nz, nw), order="F")
        tb = np.empty((nz, nw), order="F")
        self._L.radtran_opr_get(self._ptr, _d(tau), _d(w0), _d(g), _d(tb), self._err)
        self._check()
        return tau, w0, g, tb

    # ------------------------------------------------------------------ Radtran.pyx surface
    def set_bolometric_flux(self, flux):
        self._L.radtran_set_bolometric_flux_wrapper(self._ptr, _f(flux))

    def bolometric_flux(self):
        v = C.c_double()
        self._L.radtran_bolometric_flux_wrapper(self._ptr, C.byref(v))
        return v.value

    def skin_temperature(self, bond_albedo):
        v = C.c_double()
        self._L.radtran_skin_temperature_wrapper(self._ptr, _f(bond_albedo), C.byref(v))
        return v.value

    def equilibrium_temperature(self, bond_albedo):
        v = C.c_double()
        self._L.radtran_equilibrium_temperature_wrapper(self._ptr, _f(bond_albedo), C.byref(v))
        return v.value

    def _vec_get(self, name, size_name=None):
        n = C.c_int()
        getattr(self._L, "radtran_%s_get_size" % (size_name or name))(self._ptr, C.byref(n))
        arr = np.empty(n.value, np.double)
        getattr(self._L, "radtran_%s_get" % name)(self._ptr, C.byref(n), _d(arr))
        return arr

    def _vec_set(self, name, arr, size_name=None):
        arr = _c(arr)
        n = C.c_int()
        getattr(self._L, "radtran_%s_get_size" % (size_name or name))(self._ptr, C.byref(n))
        if arr.ndim != 1 or arr.shape[0] != n.value:
            raise ClimaException('"%s" is the wrong size' % name)
        getattr(self._L, "radtran_%s_set" % name)(self._ptr, C.byref(n), _d(arr))

    zenith_u = property(lambda s: s._vec_get("zenith_u"), lambda s, a: s._vec_set("zenith_u", a),
                        doc="cosine of the zenith angles")
    zenith_weights = property(lambda s: s._vec_get("zenith_weights", "zenith_u"),
                              lambda s, a: s._vec_set("zenith_weights", a, "zenith_u"))
    surface_albedo = property(lambda s: s._vec_get("surface_albedo"), lambda s, a: s._vec_set("surface_albedo", a),
                              doc="surface albedo in each solar bin")
    surface_emissivity = property(lambda s: s._vec_get("surface_emissivity"),
                                  lambda s, a: s._vec_set("surface_emissivity", a),
                                  doc="surface emissivity in each IR bin")
    photons_sol = property(lambda s: s._vec_get("photons_sol"))
    f_total = property(lambda s: s._vec_get("f_total"))

    def _scalar(name, ctype):
        def get(self):
            v = ctype()
            getattr(self._L, "radtran_%s_get" % name)(self._ptr, C.byref(v))
            return bool(v.value) if ctype is C.c_bool else v.value

        def set_(self, val):
            getattr(self._L, "radtran_%s_set" % name)(self._ptr, C.byref(ctype(bool(val) if ctype is C.c_bool else float(val))))

        return property(get, set_)

    has_hard_surface = _scalar("has_hard_surface", C.c_bool)  # logical(c_bool), clima/cython/Radtran_pxd.pxd:45-46
    photon_scale_factor = _scalar("photon_scale_factor", C.c_double)
    ir_tau_min = _scalar("ir_tau_min", C.c_double)
    diurnal_fac = _scalar("diurnal_fac", C.c_double)
    del _scalar

    def _sub(self, name, cls):
        p = C.c_void_p()
        getattr(self._L, "radtran_%s_get" % name)(self._ptr, C.byref(p))
        obj = cls(self._L, p)
        obj._parent = self  # keep the handle alive while the borrowed pointer is in use
        return obj

    ir = property(lambda s: s._sub("ir", RTChannel), doc="RTChannel of the IR bins")
    sol = property(lambda s: s._sub("sol", RTChannel), doc="RTChannel of the solar bins")
    wrk_ir = property(lambda s: s._sub("wrk_ir", ClimaRadtranWrk), doc="IR results")
    wrk_sol = property(lambda s: s._sub("wrk_sol", ClimaRadtranWrk), doc="solar results")
