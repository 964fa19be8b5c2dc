"""The C-ABI library loads and exports every symbol include/clima_radtran_hip.h declares
(no compute calls: these run without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "clima_radtran_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bvoid\s+([a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(hip_lib):
    names = _declared_symbols()
    assert len(names) > 60
    for n in names:
        assert hasattr(hip_lib, n), "library does not export %s" % n


def test_python_signature_table_matches_header(hip_lib):
    from clima_amd import lib
    assert sorted(lib.SIGNATURES) == _declared_symbols()


def test_handle_lifecycle_and_argument_validation(hip_lib):
    """Host-side logic that needs no device: construction-time checks and their messages."""
    L = hip_lib
    err = C.create_string_buffer(1025)
    h = C.c_void_p()
    L.allocate_radtran(C.byref(h))
    assert h.value

    def i(v):
        return C.byref(C.c_int(v))

    wavl = (C.c_double * 4)(100.0, 200.0, 400.0, 800.0)
    L.radtran_create_begin(h, i(0), i(2), i(0), i(3), wavl, err)
    assert err.value == b'"nz" can not be less than 1.'            # clima_radtran.f90:149-152
    L.radtran_create_begin(h, i(4), i(2), i(0), i(3), wavl, err)
    assert err.value == b""
    bad = (C.c_double * 3)(100.0, 210.0, 400.0)
    ok = (C.c_double * 3)(200.0, 400.0, 800.0)
    L.radtran_set_channels(h, i(3), bad, i(3), ok, err)
    assert b"not compatible with the k-distribution wavelength bins" in err.value  # types_create.f90:253-261
    L.radtran_set_channels(h, i(3), ok, i(3), (C.c_double * 3)(100.0, 200.0, 400.0), err)
    assert err.value == b""
    L.radtran_create_end(h, i(1), C.byref(C.c_double(0.3)), err)
    assert b"There are no k-distributions" in err.value            # clima_radtran_types.f90:594-597
    # a handle that was never constructed refuses to run
    L.radtran_radiate_resident(h, i(1), i(1), err)
    assert b"not constructed" in err.value
    L.deallocate_radtran(h)
    L.radtran_synchronize(None, err)                                # null handle is rejected
    assert b"invalid Radtran handle" in err.value


def test_has_hard_surface_is_a_one_byte_bool(hip_lib):
    """`logical(c_bool)` / `bool*` in the reference (clima/fortran/Radtran.f90:211-227,
    clima/cython/Radtran_pxd.pxd:45-46: Radtran.pyx passes the address of a 1-byte local).  The getter
    must write exactly one byte, the setter read exactly one: the slot sits between guard bytes."""
    L = hip_lib
    h = C.c_void_p()
    L.allocate_radtran(C.byref(h))
    buf = (C.c_ubyte * 9)(*([0xAB] * 9))
    slot = C.cast(C.byref(buf, 4), C.POINTER(C.c_bool))
    L.radtran_has_hard_surface_get(h, slot)              # default .true. (clima_radtran.f90:60)
    assert list(buf) == [0xAB] * 4 + [1] + [0xAB] * 4
    buf[4] = 0                                            # False next to non-zero neighbours
    L.radtran_has_hard_surface_set(h, slot)
    out = (C.c_ubyte * 9)(*([0xCD] * 9))
    L.radtran_has_hard_surface_get(h, C.cast(C.byref(out, 4), C.POINTER(C.c_bool)))
    assert list(out) == [0xCD] * 4 + [0] + [0xCD] * 4    # 4 bytes read as an int would have said True
    buf[4] = 1
    L.radtran_has_hard_surface_set(h, slot)
    L.radtran_has_hard_surface_get(h, C.cast(C.byref(out, 4), C.POINTER(C.c_bool)))
    assert list(out) == [0xCD] * 4 + [1] + [0xCD] * 4
    L.deallocate_radtran(h)


def test_header_declares_bool_for_has_hard_surface():
    text = open(os.path.join(ROOT, "include", "clima_radtran_hip.h")).read()
    assert "radtran_has_hard_surface_get(void *ptr, bool *val)" in text
    assert "radtran_has_hard_surface_set(void *ptr, const bool *val)" in text
    from clima_amd import lib
    assert lib.SIGNATURES["radtran_has_hard_surface_get"][1] is C.POINTER(C.c_bool)


def test_no_cpu_fallback_in_product():
    """The product path must not reach into oracle/ (parity claims depend on it)."""
    pkg = os.path.join(ROOT, "clima_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".f90")):
                src = open(os.path.join(base, f), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "liborc" not in src, f


def test_opacities2yaml_text(hip_lib):
    """radtran_opacities2yaml_wrapper_{1,2} (clima/fortran/Radtran.f90:41-75) print what
    OpticalProperties_opacities2yaml prints (clima_radtran_types.f90:328-430); the tables and names
    are host-side state, so this needs no device."""
    import numpy as np
    L = hip_lib
    err = C.create_string_buffer(1025)
    h = C.c_void_p()
    L.allocate_radtran(C.byref(h))
    dp = C.POINTER(C.c_double)

    def i(v):
        return C.byref(C.c_int(v))

    def d(a):
        return a.ctypes.data_as(dp)

    nw, ng, nP, nT = 4, 8, 3, 3
    wavl = np.linspace(100.0, 500.0, nw + 1)
    L.radtran_create_begin(h, i(5), i(3), i(1), i(nw), d(wavl), err)
    assert err.value == b""
    w = np.full(ng, 1.0 / ng)
    lp, tt, kk = np.linspace(-3, 0, nP), np.linspace(100.0, 300.0, nT), np.zeros(nw * nT * nP * ng)
    for sp in (1, 3):
        L.radtran_add_ktable(h, i(sp), i(ng), d(w), i(nP), d(lp), i(nT), d(tt), d(kk), err)
        assert err.value == b""
    xs1 = np.zeros(nw * nT)
    L.radtran_add_xsection(h, i(0), i(1), i(2), i(2), i(nT), d(tt), d(xs1), err)   # CIA N2-N2
    L.radtran_add_xsection(h, i(0), i(1), i(3), i(2), i(nT), d(tt), d(xs1), err)   # CIA CO2-N2
    xs0 = np.zeros(nw)
    L.radtran_add_xsection(h, i(1), i(0), i(3), i(0), i(0), d(tt), d(xs0), err)    # Rayleigh CO2
    L.radtran_add_xsection(h, i(3), i(0), i(1), i(0), i(0), d(tt), d(xs0), err)    # photolysis H2O
    L.radtran_set_water_continuum(h, i(1), i(nT), d(tt), d(xs1), d(xs1), err)
    rad = np.array([1e-6, 1e-4])
    pt = np.zeros(nw * 2)
    L.radtran_add_particle(h, i(1), i(2), d(rad), d(pt), d(pt), d(pt), err)
    assert err.value == b""
    L.radtran_set_names(h, b"H2O\nN2\nCO2", b"HCaer1", err)
    assert err.value == b""
    L.radtran_set_opacity_labels(h, b"RandomOverlapResortRebin", b"MT_CKD", b"khare1984", err)
    n, cp = C.c_int(), C.c_void_p()
    L.radtran_opacities2yaml_wrapper_1(h, C.byref(n), C.byref(cp))
    buf = C.create_string_buffer(n.value + 1)
    L.radtran_opacities2yaml_wrapper_2(h, C.byref(cp), C.byref(n), buf)
    assert buf.value.decode() == (
        "  k-method: RandomOverlapResortRebin\n"
        "  opacities:\n"
        "    k-distributions: [H2O, CO2]\n"
        "    CIA: [N2-N2, CO2-N2]\n"
        "    rayleigh: [CO2]\n"
        "    photolysis-xs: [H2O]\n"
        "    water-continuum: MT_CKD\n"
        "    particle-xs: [{name: HCaer1, data: khare1984}]")
    L.radtran_set_names(h, b"H2O\nN2", b"HCaer1", err)
    assert b"does not match" in err.value
    L.deallocate_radtran(h)
