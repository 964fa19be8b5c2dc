"""Construction from files behind the C ABI (radtran_create_from_files / radtran_load_from_files,
clima_amd/csrc/radtran_loader.hip): the reference's constructor `Radtran(settings_f, star_f, num_zenith_angles,
surface_albedo, nz, datadir, err)` (/root/reference/src/radtran/clima_radtran.f90:98-126, loaders
src/radtran/clima_radtran_types_create.f90) for hosts that do not link the reference's own loaders.

No device is needed to check the loader: radtran_load_from_files leaves the handle in its "begun" state with every
table handed over, and clima_test_host_tables_digest hashes those host-side tables (FNV-1a over metadata and float64
bytes).  The C++ loader must produce EXACTLY the tables the Python loader (clima_amd/data_loader.py) hands over for the
same files -- digest for digest -- on the committed data directory (written by the HDF5 C library: chunked, deflated,
partly float32) and on directories written here, and refuse what the reference refuses with the reference's texts."""
import ctypes as C
import os
import shutil
import subprocess
import struct

import numpy as np
import pytest

from clima_amd import synthetic as S

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
DATADIR_C = os.path.join(ROOT, "tests", "golden", "datadir_c")


def _fnv(h, b):
    for byte in b:
        h ^= byte
        h = (h * 1099511628211) & 0xffffffffffffffff
    return h


def _ints(h, *v):
    return _fnv(h, struct.pack("<%di" % len(v), *v))


def _vals(h, a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return _fnv(h, a.tobytes()) if a.size else h


GROUPS = ("extents + grid", "k-tables", "CIA", "Rayleigh", "absorption / photolysis", "continuum", "particles", "channels + photons")


def python_digest(t, nz):
    """The digests of clima_test_host_tables_digest computed from a LoadedTables."""
    H0 = 1469598103934665603
    out = []
    h = _ints(H0, nz, t.nsp, t.np_, t.nw)
    out.append(_vals(h, t.wavl))
    h = H0
    for k in t.ktables:
        h = _ints(h, k["sp_ind"], len(k["weights"]), len(k["log10P"]), len(k["temp"]))
        for name in ("weights", "log10P", "temp", "log10k"):
            h = _vals(h, k[name])
    out.append(h)
    # the library keeps CIA (0), Rayleigh (1) and absorption / photolysis (2, 3) entries in separate lists
    for types in ((0,), (1,), (2, 3)):
        h = H0
        for x in t.xsections:
            if x["xs_type"] in types:
                temp = x.get("temp")
                h = _ints(h, x["xs_type"], x["dim"], x["sp1"], x.get("sp2", -1), 0 if temp is None else len(temp))
                if temp is not None:
                    h = _vals(h, temp)
                h = _vals(h, x["data"])
        out.append(h)
    c = t.continuum
    h = _ints(H0, 1 if c is not None else 0, c["LH2O"] if c is not None else -1, len(c["temp"]) if c is not None else 0)
    if c is not None:
        for name in ("temp", "log10_H2O", "log10_foreign"):
            h = _vals(h, c[name])
    out.append(h)
    h = H0
    for p in t.particles:
        h = _ints(h, p["p_ind"], len(p["radii"]))
        for name in ("radii", "w0", "qext", "gt"):
            h = _vals(h, p[name])
    out.append(h)
    h = H0
    for a in (t.ir_wavl, t.sol_wavl, t.photons_sol):
        h = _vals(h, a)
    out.append(h)
    return out


def differing(a, b):
    return [g for g, x, y in zip(GROUPS, a, b) if x != y]


def c_load(L, settings, star, nz, datadir):
    h = C.c_void_p()
    L.allocate_radtran(C.byref(h))
    err = C.create_string_buffer(1025)
    L.radtran_load_from_files(h, settings.encode(), star.encode(), C.byref(C.c_int(nz)), datadir.encode(), err)
    d = (C.c_ulonglong * 8)()
    L.clima_test_host_tables_digest(h, d)
    L.deallocate_radtran(h)
    return err.value.decode(), list(d)


def test_committed_directory_loads_to_the_python_loaders_tables(hip_lib):
    from clima_amd import data_loader as D
    settings, star = os.path.join(DATADIR_C, "settings.yaml"), os.path.join(DATADIR_C, "star.txt")
    t = D.load_tables(settings, star, DATADIR_C)
    err, dig = c_load(hip_lib, settings, star, 50, DATADIR_C)
    assert err == ""
    assert differing(dig, python_digest(t, 50)) == []
    # (and the digests do see the tables: one value changed changes its group's)
    t.photons_sol[3] = np.nextafter(t.photons_sol[3], 1.0)
    assert differing(dig, python_digest(t, 50)) == ["channels + photons"]


@pytest.fixture(scope="module")
def written(tmp_path_factory):
    from tests.datadir_fixture import write_datadir
    root = str(tmp_path_factory.mktemp("datadir_cabi"))
    tb = S.make_tables(nw=24, nT=6, nP=5, seed=41)
    write_datadir(root, tb)
    return root, tb


SETTINGS_VARIANTS = {
    # everything the directory holds, flow style as the reference's templates write it
    "on": """
optical-properties:
  species:
    gases: [H2O, CO2, O2, N2, O3, CH4]
    particles: [HCaer1]
  k-method: RandomOverlapResortRebin
  opacities: {k-distributions: true, CIA: true, rayleigh: true, photolysis-xs: true,
    water-continuum: MT_CKD, particle-xs: [{name: HCaer1, data: khare1984}]}
""",
    # block style, explicit lists, comments, quoted strings, no continuum (so H2O pairs would be legal), no particles
    "lists": """
# a comment line
planet:
  surface-albedo: 0.3   # not ours
optical-properties:
  species:
    gases:
      - H2O
      - "CO2"
      - O2
      - N2
      - O3
      - CH4
  k-method: 'RandomOverlapResortRebin'
  opacities:
    k-distributions: [CO2, H2O]
    CIA:
      - N2-N2
      - CO2-CO2
    rayleigh: [N2, O2]
    photolysis-xs:
      - O3
""",
    "konly": """
optical-properties:
  species: {gases: [H2O, CO2, O2, N2, O3, CH4]}
  k-method: RandomOverlapResortRebin
  opacities: {k-distributions: [O3], CIA: off, rayleigh: false}
""",
}


@pytest.mark.parametrize("name", sorted(SETTINGS_VARIANTS))
def test_settings_styles_and_selections(hip_lib, written, tmp_path, name):
    from clima_amd import data_loader as D
    root, tb = written
    settings = str(tmp_path / "settings.yaml")
    with open(settings, "w") as f:
        f.write(SETTINGS_VARIANTS[name])
    star = os.path.join(root, "star.txt")
    t = D.load_tables(settings, star, root)
    err, dig = c_load(hip_lib, settings, star, 30, root)
    assert err == ""
    assert differing(dig, python_digest(t, 30)) == [], name


def test_reference_error_texts_from_the_c_loader(hip_lib, written, tmp_path):
    root, tb = written
    star = os.path.join(root, "star.txt")

    def run(text, star_file=star, datadir=root):
        settings = str(tmp_path / "s.yaml")
        with open(settings, "w") as f:
            f.write(text)
        return c_load(hip_lib, settings, star_file, 30, datadir)[0]

    head = "optical-properties:\n  species: {gases: [H2O, CO2, O2, N2, O3, CH4], particles: [HCaer1]}\n"
    km = "  k-method: RandomOverlapResortRebin\n"
    assert run(head + "  k-method: Foo\n  opacities: {k-distributions: true}\n").startswith('k-method "Foo" in "')
    assert run(head + km + "  opacities: {k-distributions: [H2O, CO2, H2O]}\n") == '"H2O" is a duplicate in k-distributions'
    assert run(head + km + "  opacities: {CIA: true}\n") == "You must specify at least one k-distribution in the settings file."
    assert run(head + km + "  opacities: {k-distributions: [Xe]}\n") == \
        'Species "Xe" in optical property "k-distributions" is not in the list of species.'
    assert run(head + km + "  opacities: {k-distributions: [H2O], CIA: [N2-Xe]}\n") == \
        'Could not parse CIA species pair "N2-Xe" into two known species.'
    assert run(head + km + "  opacities: {k-distributions: [H2O], water-continuum: CKD9}\n") == 'Continuum "CKD9" is not avaliable.'
    assert run(head + km + "  opacities: {k-distributions: [H2O], particle-xs: [{name: Soot, data: khare1984}]}\n") == \
        'Species "Soot" in optical property "particle-xs" is not in the list of particles.'
    assert run("planet: {}\n").endswith('"optical-properties" is required')
    assert run(head + km + "  opacities: {k-distributions: [H2O]}\n", star_file=str(tmp_path / "nostar.txt")).endswith("does not exist.")
    shutil.copy(os.path.join(root, "CIA", "N2-N2.h5"), os.path.join(root, "CIA", "H2O-H2O.h5"))
    try:
        assert "double count opacity" in run(head + km + "  opacities: {k-distributions: [H2O], CIA: [H2O-H2O], water-continuum: MT_CKD}\n")
    finally:
        os.remove(os.path.join(root, "CIA", "H2O-H2O.h5"))
    # a file that is not HDF5 where a k-table should be
    bad = str(tmp_path / "bad")
    shutil.copytree(root, bad)
    with open(os.path.join(bad, "kdistributions", "H2O.h5"), "w") as f:
        f.write("not hdf5")
    assert run(head + km + "  opacities: {k-distributions: [H2O]}\n", datadir=bad).startswith('Failed to read "')


@pytest.mark.gpu
def test_constructed_from_files_behind_the_c_abi_is_the_python_constructed_handle():
    """On the device: the handle radtran_create_from_files builds gives, call for call, bit for bit what
    Radtran.from_files (Python loader + radtran_create_*) gives, and prints the same opacities2yaml."""
    from clima_amd import lib as _lib
    from clima_amd.radtran import Radtran
    settings, star = os.path.join(DATADIR_C, "settings.yaml"), os.path.join(DATADIR_C, "star.txt")
    nz = 50
    a = Radtran.from_files(settings, star, 4, 0.3, nz, DATADIR_C)
    b = Radtran.from_files_c(settings, star, 4, 0.3, nz, DATADIR_C)
    col = S.modern_earth_column(nz)
    assert a.TOA_fluxes(*col.args()) == b.TOA_fluxes(*col.args())
    np.testing.assert_array_equal(np.asarray(a.f_total), np.asarray(b.f_total))
    for x, y in zip(a.opr(), b.opr()):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(np.asarray(a.wrk_sol.amean), np.asarray(b.wrk_sol.amean))
    assert a.opacities2yaml() == b.opacities2yaml()
    assert _lib is not None


def _fortran_from_files():
    from clima_amd import build
    build.build()
    if build.build_fortran_shim() is None:
        pytest.skip("amdflang is not available on this box")
    return build.FORTRAN_FROM_FILES


def _write_column(path, col, n_particles):
    with open(path, "wb") as f:
        for a in ([col["T_surface"]], col["T"], col["P"], np.asfortranarray(col["densities"]).T, col["dz"]):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        if n_particles:
            f.write(np.ascontiguousarray(np.asfortranarray(col["pdensities"]).T, dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(np.asfortranarray(col["radii"]).T, dtype="<f8").tobytes())


def test_fortran_constructor_reports_the_references_errors(tmp_path):
    """`rad = Radtran(settings_f, star_f, ...)` in the Fortran module: a refused settings file comes back as the allocated
    `err` with the reference's text (no device needed: the files are read before anything is uploaded)."""
    exe = _fortran_from_files()
    settings = str(tmp_path / "s.yaml")
    with open(settings, "w") as f:
        f.write("optical-properties:\n  species: {gases: [H2O, CO2]}\n  k-method: Foo\n  opacities: {k-distributions: true}\n")
    col = str(tmp_path / "col.bin")
    open(col, "wb").close()
    r = subprocess.run([exe, settings, os.path.join(DATADIR_C, "star.txt"), DATADIR_C, "50", "4", "0.3", col, str(tmp_path / "o.txt")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert 'error: k-method "Foo" in "%s" is not an option.' % settings in r.stdout


@pytest.mark.gpu
def test_fortran_host_constructed_from_files_matches_the_python_handle(tmp_path):
    """A Fortran program shaped like the reference's tests/test_radtran.f90 that builds its object with the reference's
    constructor call -- `rad = Radtran(settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir, err)` -- and no
    Python in the loop, against Radtran.from_files: same bits."""
    from clima_amd.radtran import Radtran
    exe = _fortran_from_files()
    settings, star = os.path.join(DATADIR_C, "settings.yaml"), os.path.join(DATADIR_C, "star.txt")
    nz, nzen, albedo = 50, 4, 0.3
    col = S.modern_earth_column(nz)
    a = Radtran.from_files(settings, star, nzen, albedo, nz, DATADIR_C)
    colf, res = str(tmp_path / "col.bin"), str(tmp_path / "res.txt")
    _write_column(colf, col, a.np)
    out = subprocess.run([exe, settings, star, DATADIR_C, str(nz), str(nzen), repr(albedo), colf, res], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    isr, olr = a.TOA_fluxes(*col.args())
    vals = np.array(open(res).read().split(), dtype=float)
    nw_ir, nw_sol = len(a.ir.freq) - 1, len(a.sol.freq) - 1
    assert (vals[0], vals[1]) == (isr, olr)
    p = 2
    for want in (a.wrk_ir.fup_n, a.wrk_sol.fdn_n, a.f_total, np.asarray(a.wrk_ir.fup_a)[nz, :], np.asarray(a.wrk_sol.amean)[0, :],
                 a.photons_sol):
        want = np.asarray(want)
        np.testing.assert_array_equal(vals[p:p + want.size], want)
        p += want.size
    assert p == len(vals) and nw_ir and nw_sol
    assert a.opacities2yaml().strip() in out.stdout
