"""ctypes bindings for the parity oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product (clima_amd/) never does.

Two libraries are wrapped:
  * oracle/liborc.so                      -- plain-C restatement (clima_oracle.c)
  * oracle/_ref/libclima_twostream_ref.so -- the reference's own two-stream solver
    (src/radtran/clima_radtran_twostream.f90 compiled unmodified with amdflang; the
    flang-mangled module-procedure symbols are called directly).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ERR_LEN = 1024
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

XS_CIA, XS_RAYLEIGH, XS_ABSORPTION, XS_PHOTOLYSIS = 0, 1, 2, 3


def build(force=False):
    """Compile liborc.so (and _ref when /root/reference is present)."""
    src = os.path.join(_HERE, "clima_oracle.c")
    for name in ("liborc.so", "liborc_fma.so"):
        lib = os.path.join(_HERE, name)
        stale = (not os.path.exists(lib)) or os.path.getmtime(lib) < os.path.getmtime(src)
        if force or stale:
            subprocess.check_call(["make", "-C", _HERE, name], stdout=subprocess.DEVNULL)
    ref = os.path.join(_HERE, "_ref", "libclima_twostream_ref.so")
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(ref)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _arr(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _farr(a):
    """(nz, n) array -> column-major buffer."""
    a = np.asfortranarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


_libs = {}


def lib(variant=""):
    """variant "" = liborc.so (no FMA contraction, the oracle proper); "fma" = liborc_fma.so,
    the same source compiled with contraction (conditioning yardstick only)."""
    if variant not in _libs:
        build()
        # CLIMA_ORACLE_DIR: another build of the same libraries (the sanitizer build of tools/sanitize.sh)
        L = C.CDLL(os.path.join(os.environ.get("CLIMA_ORACLE_DIR") or _HERE, "liborc%s.so" % ("_" + variant if variant else "")))
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_add_ktable.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, _dp, C.c_char_p]
        L.orc_add_xsection.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_char_p]
        L.orc_set_water_continuum.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, C.c_char_p]
        L.orc_add_particle.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_char_p]
        L.orc_set_channels.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _dp, C.c_char_p]
        L.orc_set_photons_sol.argtypes = [C.c_void_p, C.c_int, _dp, C.c_char_p]
        L.orc_set_custom_optical_properties.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, _dp,
                                                        C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_char_p]
        L.orc_unset_custom_optical_properties.argtypes = [C.c_void_p]
        L.orc_unset_custom_optical_properties.restype = None
        L.orc_finalize.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_char_p]
        L.orc_set_zenith.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.orc_get_zenith.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_set_surface_albedo.argtypes = [C.c_void_p, _dp]
        L.orc_set_surface_emissivity.argtypes = [C.c_void_p, _dp]
        L.orc_set_scalars.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double]
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_get_max_threads.restype = C.c_int
        L.orc_radiate.argtypes = [C.c_void_p, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_char_p]
        L.orc_toa_fluxes.argtypes = [C.c_void_p, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int, _dp, _dp, C.c_char_p]
        L.orc_dims.argtypes = [C.c_void_p] + [_ip] * 7
        L.orc_get_wrk.argtypes = [C.c_void_p, C.c_int] + [_dp] * 6
        L.orc_get_f_total.argtypes = [C.c_void_p, _dp]
        L.orc_get_opr.argtypes = [C.c_void_p] + [_dp] * 4
        L.orc_get_channel.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.orc_two_stream_solar.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp]
        L.orc_two_stream_ir.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, C.c_int, C.c_double, _dp, _dp, _dp]
        L.orc_tridiag.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
        L.orc_planck_fcn.restype = C.c_double
        L.orc_planck_fcn.argtypes = [C.c_double, C.c_double]
        L.orc_ten2power.restype = C.c_double
        L.orc_ten2power.argtypes = [C.c_double]
        L.orc_interp1d.restype = C.c_double
        L.orc_interp1d.argtypes = [C.c_int, _dp, _dp, C.c_double]
        L.orc_interp2d.restype = C.c_double
        L.orc_interp2d.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double]
        L.orc_mrgrnk.argtypes = [C.c_int, _dp, _ip]
        L.orc_rebin.argtypes = [C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        L.orc_gauss_legendre.argtypes = [C.c_int, _dp, _dp]
        _libs[variant] = L
    return _libs[variant]


class OracleError(Exception):
    pass


class _Wrk:
    pass


class OracleRadtran:
    """CPU oracle with the shape of clima.Radtran (clima/cython/Radtran.pyx)."""

    def __init__(self, tables, nz, num_zenith_angles, surface_albedo, variant=""):
        """tables: clima_amd.synthetic.TableSet-like object (duck-typed)."""
        L = lib(variant)
        self._L = L
        t = tables
        self.nz, self.nsp, self.np_ = nz, t.nsp, t.np_
        err = C.create_string_buffer(ERR_LEN + 1)
        _, p = _arr(t.wavl)
        self._h = L.orc_create(nz, t.nsp, t.np_, t.nw, p)

        def chk(rc):
            if rc:
                raise OracleError(err.value.decode())

        for k in t.ktables:
            a = [_arr(k["weights"]), _arr(k["log10P"]), _arr(k["temp"]), _arr(k["log10k"])]
            chk(L.orc_add_ktable(self._h, k["sp_ind"], len(k["weights"]), a[0][1], len(k["log10P"]), a[1][1],
                                 len(k["temp"]), a[2][1], a[3][1], err))
        for x in t.xsections:
            temp = x.get("temp")
            ta = _arr(temp if temp is not None else np.zeros(1))
            da = _arr(x["data"])
            chk(L.orc_add_xsection(self._h, x["xs_type"], x["dim"], x["sp1"], x.get("sp2", -1),
                                   0 if temp is None else len(temp), ta[1], da[1], err))
        if t.continuum is not None:
            c = t.continuum
            a = [_arr(c["temp"]), _arr(c["log10_H2O"]), _arr(c["log10_foreign"])]
            chk(L.orc_set_water_continuum(self._h, c["LH2O"], len(c["temp"]), a[0][1], a[1][1], a[2][1], err))
        for p_ in t.particles:
            a = [_arr(p_["radii"]), _arr(p_["w0"]), _arr(p_["qext"]), _arr(p_["gt"])]
            chk(L.orc_add_particle(self._h, p_["p_ind"], len(p_["radii"]), a[0][1], a[1][1], a[2][1], a[3][1], err))
        a = [_arr(t.ir_wavl), _arr(t.sol_wavl)]
        chk(L.orc_set_channels(self._h, len(t.ir_wavl), a[0][1], len(t.sol_wavl), a[1][1], err))
        a = _arr(t.photons_sol)
        chk(L.orc_set_photons_sol(self._h, len(t.photons_sol), a[1], err))
        chk(L.orc_finalize(self._h, num_zenith_angles, surface_albedo, err))
        d = [C.c_int() for _ in range(7)]
        L.orc_dims(self._h, *[C.byref(x) for x in d])
        self.nw, self.ng, self.nw_ir, self.nw_sol = d[1].value, d[2].value, d[3].value, d[4].value
        self.ir_start, self.sol_start = d[5].value, d[6].value
        self.nzen = num_zenith_angles
        self.diurnal_fac, self.has_hard_surface, self.ir_tau_min, self.photon_scale_factor = 0.5, True, 1e-6, 1.0

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_destroy(self._h)
            self._h = None

    # ---- fields
    def set_scalars(self, diurnal_fac=None, has_hard_surface=None, ir_tau_min=None, photon_scale_factor=None):
        if diurnal_fac is not None: self.diurnal_fac = diurnal_fac
        if has_hard_surface is not None: self.has_hard_surface = has_hard_surface
        if ir_tau_min is not None: self.ir_tau_min = ir_tau_min
        if photon_scale_factor is not None: self.photon_scale_factor = photon_scale_factor
        self._L.orc_set_scalars(self._h, self.diurnal_fac, int(self.has_hard_surface), self.ir_tau_min,
                                self.photon_scale_factor)

    def set_custom_optical_properties(self, wv, P, dtau_dz, w0, g0):
        """clima_radtran.f90:494-506.  Arrays (size(P), size(wv)); raises OracleError with the
        reference's message."""
        err = C.create_string_buffer(1025)
        a = [_arr(wv), _arr(P)] + [_arr(np.asfortranarray(x, dtype=float).ravel(order="F")) for x in (dtau_dz, w0, g0)]
        sh = [np.shape(x) for x in (dtau_dz, w0, g0)]
        rc = self._L.orc_set_custom_optical_properties(self._h, len(a[0][0]), a[0][1], len(a[1][0]), a[1][1],
                                                       sh[0][0], sh[0][1], a[2][1], sh[1][0], sh[1][1], a[3][1],
                                                       sh[2][0], sh[2][1], a[4][1], err)
        if rc != 0:
            raise OracleError(err.value.decode())

    def unset_custom_optical_properties(self):
        self._L.orc_unset_custom_optical_properties(self._h)

    def set_zenith(self, u, w):
        ua, wa = _arr(u), _arr(w)
        self.nzen = len(ua[0])
        self._L.orc_set_zenith(self._h, self.nzen, ua[1], wa[1])

    def get_zenith(self):
        u, w = np.empty(self.nzen), np.empty(self.nzen)
        self._L.orc_get_zenith(self._h, u.ctypes.data_as(_dp), w.ctypes.data_as(_dp))
        return u, w

    def set_surface_albedo(self, a):
        a = _arr(np.broadcast_to(a, (self.nw_sol,)))
        self._L.orc_set_surface_albedo(self._h, a[1])

    def set_surface_emissivity(self, e):
        e = _arr(np.broadcast_to(e, (self.nw_ir,)))
        self._L.orc_set_surface_emissivity(self._h, e[1])

    # ---- the path
    def radiate(self, T_surface, T, P, densities, dz, pdensities=None, radii=None, compute_solar=True,
                compute_opacity=True):
        err = C.create_string_buffer(ERR_LEN + 1)
        a = [_arr(T), _arr(P), _farr(densities), _arr(dz)]
        pd = _farr(pdensities) if pdensities is not None else (None, None)
        ra = _farr(radii) if radii is not None else (None, None)
        rc = self._L.orc_radiate(self._h, T_surface, a[0][1], a[1][1], a[2][1], a[3][1], pd[1], ra[1],
                                 int(compute_solar), int(compute_opacity), err)
        if rc:
            raise OracleError(err.value.decode())

    def TOA_fluxes(self, *args, **kw):
        self.radiate(*args, **kw)
        ir, sol = self.wrk(0), self.wrk(1)
        nz = self.nz
        return sol.fdn_n[nz] - sol.fup_n[nz], -(ir.fdn_n[nz] - ir.fup_n[nz])

    def wrk(self, which):
        nwc = self.nw_sol if which else self.nw_ir
        nz = self.nz
        w = _Wrk()
        w.fup_a = np.empty((nz + 1, nwc), order="F")
        w.fdn_a = np.empty((nz + 1, nwc), order="F")
        w.amean = np.empty((nz + 1, nwc), order="F")
        w.tau_band = np.empty((nz, nwc), order="F")
        w.fup_n = np.empty(nz + 1)
        w.fdn_n = np.empty(nz + 1)
        self._L.orc_get_wrk(self._h, which, *[x.ctypes.data_as(_dp) for x in
                                              (w.fup_a, w.fdn_a, w.fup_n, w.fdn_n, w.amean, w.tau_band)])
        return w

    @property
    def wrk_ir(self):
        return self.wrk(0)

    @property
    def wrk_sol(self):
        return self.wrk(1)

    @property
    def f_total(self):
        f = np.empty(self.nz + 1)
        self._L.orc_get_f_total(self._h, f.ctypes.data_as(_dp))
        return f

    def opr(self):
        """OpticalPropertiesResult: tau,w0 (nz,ng,nw) F-order TOA-first; g,tau_band (nz,nw)."""
        nz, ng, nw = self.nz, self.ng, self.nw
        tau = np.empty((nz, ng, nw), order="F")
        w0 = np.empty((nz, ng, nw), order="F")
        g = np.empty((nz, nw), order="F")
        tb = np.empty((nz, nw), order="F")
        self._L.orc_get_opr(self._h, *[x.ctypes.data_as(_dp) for x in (tau, w0, g, tb)])
        return tau, w0, g, tb


# ---------------------------------------------------------------- unit-level wrappers

def two_stream_ir(tau, w0, gt, emissivity, has_hard_surface, tau_min, bplanck, variant=""):
    nz = len(tau)
    a = [_arr(tau), _arr(w0), _arr(gt), _arr(bplanck)]
    fup, fdn = np.empty(nz + 1), np.empty(nz + 1)
    lib(variant).orc_two_stream_ir(nz, a[0][1], a[1][1], a[2][1], emissivity, int(has_hard_surface), tau_min, a[3][1],
                            fup.ctypes.data_as(_dp), fdn.ctypes.data_as(_dp))
    return fup, fdn


def two_stream_solar(tau, w0, gt, u0, Rsfc, variant=""):
    nz = len(tau)
    a = [_arr(tau), _arr(w0), _arr(gt)]
    amean, fup, fdn = np.empty(nz + 1), np.empty(nz + 1), np.empty(nz + 1)
    sr = C.c_double()
    lib(variant).orc_two_stream_solar(nz, a[0][1], a[1][1], a[2][1], u0, Rsfc, amean.ctypes.data_as(_dp), C.byref(sr),
                               fup.ctypes.data_as(_dp), fdn.ctypes.data_as(_dp))
    return amean, sr.value, fup, fdn


def planck_fcn(nu, T):
    return lib().orc_planck_fcn(nu, T)


def interp1d(x, f, xv):
    a = [_arr(x), _arr(f)]
    return lib().orc_interp1d(len(a[0][0]), a[0][1], a[1][1], xv)


def interp2d(x, y, f, xv, yv):
    """f[ix, iy] (any numpy order)."""
    xa, ya = _arr(x), _arr(y)
    fa = np.asfortranarray(f, dtype=np.float64)
    return lib().orc_interp2d(len(xa[0]), len(ya[0]), xa[1], ya[1], fa.ctypes.data_as(_dp), xv, yv)


def mrgrnk(x):
    xa = _arr(x)
    r = np.empty(len(xa[0]), dtype=np.int32)
    lib().orc_mrgrnk(len(xa[0]), xa[1], r.ctypes.data_as(_ip))
    return r


def rebin(old_bins, old_vals, new_bins):
    a = [_arr(old_bins), _arr(old_vals), _arr(new_bins)]
    out = np.empty(len(a[2][0]) - 1)
    lib().orc_rebin(len(a[1][0]), a[0][1], a[1][1], len(out), a[2][1], out.ctypes.data_as(_dp))
    return out


def gauss_legendre(n):
    x, w = np.empty(n), np.empty(n)
    lib().orc_gauss_legendre(n, x.ctypes.data_as(_dp), w.ctypes.data_as(_dp))
    return x, w


# ---------------------------------------------------------------- compiled reference (_ref)

_ref = None


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libclima_twostream_ref.so"))


def ref_lib():
    """The reference's two-stream module, compiled unmodified (oracle/Makefile `ref`)."""
    global _ref
    if _ref is None:
        build()
        R = C.CDLL(os.path.join(_HERE, "_ref", "libclima_twostream_ref.so"))
        # Fortran: all arguments by reference; explicit-shape arrays are bare pointers;
        # default logical is 4 bytes (clima_radtran_twostream.f90:10-19, 156-167).
        R.ir = getattr(R, "_QMclima_radtran_twostreamPtwo_stream_ir")
        R.sol = getattr(R, "_QMclima_radtran_twostreamPtwo_stream_solar")
        R.ir.restype = None
        R.sol.restype = None
        _ref = R
    return _ref


def ref_two_stream_ir(tau, w0, gt, emissivity, has_hard_surface, tau_min, bplanck):
    R = ref_lib()
    nz = C.c_int(len(tau))
    a = [_arr(tau), _arr(w0), _arr(gt), _arr(bplanck)]
    fup, fdn = np.empty(nz.value + 1), np.empty(nz.value + 1)
    em, tm, hs = C.c_double(emissivity), C.c_double(tau_min), C.c_int(1 if has_hard_surface else 0)
    R.ir(C.byref(nz), a[0][1], a[1][1], a[2][1], C.byref(em), C.byref(hs), C.byref(tm), a[3][1],
         fup.ctypes.data_as(_dp), fdn.ctypes.data_as(_dp))
    return fup, fdn


def ref_two_stream_solar(tau, w0, gt, u0, Rsfc):
    R = ref_lib()
    nz = C.c_int(len(tau))
    a = [_arr(tau), _arr(w0), _arr(gt)]
    amean, fup, fdn = np.empty(nz.value + 1), np.empty(nz.value + 1), np.empty(nz.value + 1)
    u, rs, sr = C.c_double(u0), C.c_double(Rsfc), C.c_double()
    R.sol(C.byref(nz), a[0][1], a[1][1], a[2][1], C.byref(u), C.byref(rs), amean.ctypes.data_as(_dp),
          C.byref(sr), fup.ctypes.data_as(_dp), fdn.ctypes.data_as(_dp))
    return amean, sr.value, fup, fdn
