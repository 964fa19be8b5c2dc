"""Opacity-data loader (clima_amd/data_loader.py) against a data
directory written in the reference's schema by tests/datadir_fixture.py, and the restated
futils regridding routines against first-principles properties."""
import os

import numpy as np
import pytest

from clima_amd import synthetic as S


def _bin_average(edges, x, y):
    """Independent check of inter2: trapezoid of the linearly connected points over each bin
    (edge values by np.interp, interior points as they are)."""
    out = np.empty(len(edges) - 1)
    for i in range(len(edges) - 1):
        a, b = edges[i], edges[i + 1]
        m = (x > a) & (x < b)
        xs = np.concatenate([[a], x[m], [b]])
        ys = np.concatenate([[np.interp(a, x, y)], y[m], [np.interp(b, x, y)]])
        out[i] = np.sum(0.5 * (ys[1:] + ys[:-1]) * np.diff(xs)) / (b - a)
    return out


def test_addpnt_and_inter2_semantics():
    from clima_amd.data_loader import addpnt, inter2
    x, y = np.array([1.0, 2.0, 4.0]), np.array([10.0, 20.0, 0.0])
    x2, y2 = addpnt(x, y, 3.0, 7.0)
    assert list(x2) == [1.0, 2.0, 3.0, 4.0] and list(y2) == [10.0, 20.0, 7.0, 0.0]
    x2, y2 = addpnt(x, y, 0.5, -1.0)
    assert x2[0] == 0.5 and y2[0] == -1.0
    x2, y2 = addpnt(x, y, 9.0, 5.0)
    assert x2[-1] == 9.0 and y2[-1] == 5.0
    with pytest.raises(ValueError):
        addpnt(x, y, 2.0, 0.0)            # duplicate abscissa
    # a straight line averages to its mid-bin value; areas are conserved
    xs = np.linspace(0.0, 10.0, 23)
    ys = 3.0 - 0.25 * xs
    g = np.array([0.0, 0.7, 2.0, 2.1, 6.5, 10.0])
    out = inter2(g, xs, ys)
    np.testing.assert_allclose(out, 3.0 - 0.25 * 0.5 * (g[1:] + g[:-1]), rtol=1e-13)
    rng = np.random.default_rng(0)
    ys = rng.random(23)
    out = inter2(g, xs, ys)
    np.testing.assert_allclose(out, _bin_average(g, xs, ys), rtol=1e-12)
    np.testing.assert_allclose(np.sum(out * np.diff(g)), np.sum(0.5 * (ys[1:] + ys[:-1]) * np.diff(xs)), rtol=1e-12)
    with pytest.raises(ValueError):
        inter2(np.array([-1.0, 1.0]), xs, ys)      # data do not span the grid


def test_interp_discrete_to_bins_modes():
    from clima_amd.data_loader import interp_discrete_to_bins
    x, y = np.array([2.0, 3.0, 5.0]), np.array([1.0, 3.0, 3.0])
    bins = np.array([0.5, 1.0, 2.0, 2.5, 4.0, 6.0, 9.0])
    c = interp_discrete_to_bins(bins, x, y, "Constant")
    np.testing.assert_allclose(c, [1.0, 1.0, 1.5, (0.5 * 2.5 + 3.0) / 1.5, 3.0, 3.0], rtol=1e-13)
    f = interp_discrete_to_bins(bins, x, y, "FillValue", -7.0)
    assert f[0] == -7.0 and f[-1] == -7.0 and abs(f[2] - 1.5) < 1e-12


@pytest.fixture(scope="module")
def datadir(tmp_path_factory):
    from datadir_fixture import write_datadir
    root = str(tmp_path_factory.mktemp("clima_data"))
    tb = S.modern_earth_tables(nw=30, seed=8)
    return root, tb, write_datadir(root, tb)


def test_load_tables_round_trip(datadir):
    from clima_amd import data_loader as D
    root, tb, written = datadir
    t = D.load_tables(os.path.join(root, "settings.yaml"), os.path.join(root, "star.txt"), root)
    assert t.species_names == tuple(tb.species_names) and t.particle_names == tuple(tb.particle_names)
    np.testing.assert_allclose(t.wavl, tb.wavl, rtol=1e-15)
    assert len(t.ktables) == len(tb.ktables)
    for a, b in zip(t.ktables, tb.ktables):      # k-tables pass through byte for byte
        assert a["sp_ind"] == b["sp_ind"]
        for key in ("weights", "log10P", "temp", "log10k"):
            np.testing.assert_array_equal(a[key], b[key])
    np.testing.assert_allclose(t.ir_wavl, tb.ir_wavl, rtol=1e-15)
    np.testing.assert_allclose(t.sol_wavl, tb.sol_wavl, rtol=1e-15)
    sp = list(tb.species_names)
    pad = D.LOG10TINY

    def padded(x, y):
        xx = np.concatenate([[0.0, x[0] * (1 - 1e-4)], x, [x[-1] * (1 + 1e-4), D.HUGE]])
        return xx, np.concatenate([[pad, pad], y, [pad, pad]])

    n_cia = 0
    for x in t.xsections:
        if x["xs_type"] == D.XS_CIA:
            n_cia += 1
            xf, temp, vals = written["cia"][sp[x["sp1"]] + "-" + sp[x["sp2"]]]
            np.testing.assert_array_equal(x["temp"], temp)
            for j in range(len(temp)):
                xx, yy = padded(xf, vals[:, j])
                np.testing.assert_allclose(x["data"][:, j], _bin_average(t.wavl, xx, yy), rtol=1e-10)
        elif x["xs_type"] == D.XS_RAYLEIGH:
            from datadir_fixture import RAY_PAR
            A, B, Dl = RAY_PAR[sp[x["sp1"]]]
            np.testing.assert_allclose(x["data"], D.rayleigh_vardavas(A, B, Dl, t.wavl[:-1]), rtol=1e-14)
        elif x["xs_type"] == D.XS_PHOTOLYSIS:
            xf, xs = written["pxs"][sp[x["sp1"]]]
            xx, yy = padded(xf, np.log10(xs))
            np.testing.assert_allclose(np.log10(x["data"]), _bin_average(t.wavl, xx, yy), rtol=1e-10)
    assert n_cia == len(written["cia"]) == 6
    xf, temp, a, b = written["cont"]
    for key, ref in (("log10_H2O", a), ("log10_foreign", b)):
        for j in range(len(temp)):
            xx, yy = padded(xf, ref[:, j])
            np.testing.assert_allclose(t.continuum[key][:, j], _bin_average(t.wavl, xx, yy), rtol=1e-10)
    assert t.continuum["LH2O"] == sp.index("H2O")
    xf, rad_um, w0, qext, g0 = written["part"]["HCaer1"]
    p = t.particles[0]
    np.testing.assert_allclose(p["radii"], rad_um / 1.0e4, rtol=1e-15)
    xx = np.concatenate([[0.0], xf, [D.HUGE]])
    for j in (0, len(rad_um) // 2, len(rad_um) - 1):
        yy = np.concatenate([[qext[0, j]], qext[:, j], [qext[-1, j]]])
        np.testing.assert_allclose(p["qext"][:, j], _bin_average(t.wavl, xx, yy), rtol=1e-10)
    ws, flux = written["star"]
    xx = np.concatenate([[0.0, ws[0] * (1 - 1e-4)], ws, [ws[-1] * (1 + 1e-4), D.HUGE]])
    yy = np.concatenate([[0.0, 0.0], flux, [0.0, 0.0]])
    wav = 0.5 * (t.sol_wavl[:-1] + t.sol_wavl[1:])
    np.testing.assert_allclose(t.photons_sol, _bin_average(t.sol_wavl, xx, yy) * (wav * 1e-9 * wav / D.C_LIGHT), rtol=1e-10)


def test_c_written_directory_is_chunked_deflated_and_partly_float32():
    """tests/golden/datadir_c/ was written by tests/golden/h5pack.c (HDF5 C library: chunked layout,
    shuffle + deflate, float32 storage for one k-table and the Mie tables), not by h5lite.write."""
    import ctypes as C
    from clima_amd import h5lite
    from expected_tables import DATADIR_C
    h = h5lite.lib()
    hid = C.c_int64
    for fn, res, args in (("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]), ("H5Dopen2", hid, [hid, C.c_char_p, hid]),
                          ("H5Dget_create_plist", hid, [hid]), ("H5Pget_layout", C.c_int, [hid]),
                          ("H5Pget_nfilters", C.c_int, [hid]), ("H5Dget_type", hid, [hid]), ("H5Tget_size", C.c_size_t, [hid]),
                          ("H5Pclose", C.c_int, [hid]), ("H5Tclose", C.c_int, [hid]), ("H5Dclose", C.c_int, [hid]),
                          ("H5Fclose", C.c_int, [hid])):
        getattr(h, fn).restype, getattr(h, fn).argtypes = res, args

    def props(rel, name):
        f = h.H5Fopen(os.path.join(DATADIR_C, rel).encode(), 0, 0)
        assert f >= 0
        d = h.H5Dopen2(f, name.encode(), 0)
        pl, ty = h.H5Dget_create_plist(d), h.H5Dget_type(d)
        out = (h.H5Pget_layout(pl), h.H5Pget_nfilters(pl), h.H5Tget_size(ty))
        h.H5Pclose(pl); h.H5Tclose(ty); h.H5Dclose(d); h.H5Fclose(f)
        return out

    H5D_CHUNKED = 2
    assert props("kdistributions/H2O.h5", "log10k") == (H5D_CHUNKED, 2, 8)
    assert props("kdistributions/CO2.h5", "log10k") == (H5D_CHUNKED, 2, 4)        # float32 on disk
    assert props("aerosol_xsections/khare1984/mie_khare1984.h5", "qext") == (H5D_CHUNKED, 2, 4)
    assert props("CIA/N2-N2.h5", "log10xs")[:2] == (H5D_CHUNKED, 2)


def test_c_written_directory_against_tables_built_without_the_loader():
    """Every table the loader produces from tests/golden/datadir_c/ against tests/expected_tables.py,
    which builds the same tables from the arrays that went into the files with its own regridding
    routine (no clima_amd.data_loader, no h5lite)."""
    from clima_amd import data_loader as D
    from expected_tables import DATADIR_C, expected_tables
    e = expected_tables()
    t = D.load_tables(os.path.join(DATADIR_C, "settings.yaml"), os.path.join(DATADIR_C, "star.txt"), DATADIR_C)
    assert t.species_names == e.species_names and t.particle_names == e.particle_names
    np.testing.assert_allclose(t.wavl, e.wavl, rtol=1e-15)
    np.testing.assert_allclose(t.ir_wavl, e.ir_wavl, rtol=1e-15)
    np.testing.assert_allclose(t.sol_wavl, e.sol_wavl, rtol=1e-15)
    assert len(t.ktables) == len(e.ktables) == 5
    for a, b in zip(t.ktables, e.ktables):
        assert a["sp_ind"] == b["sp_ind"]
        for key in ("weights", "log10P", "temp", "log10k"):      # pass-through (float32 data widened exactly)
            np.testing.assert_array_equal(a[key], b[key])
    assert [(x["xs_type"], x["sp1"], x.get("sp2", -1)) for x in t.xsections] == \
           [(x["xs_type"], x["sp1"], x.get("sp2", -1)) for x in e.xsections]
    for a, b in zip(t.xsections, e.xsections):
        np.testing.assert_allclose(a["data"], b["data"], rtol=1e-11)
        if b["temp"] is not None:
            np.testing.assert_array_equal(a["temp"], b["temp"])
    for key in ("log10_H2O", "log10_foreign", "temp"):
        np.testing.assert_allclose(t.continuum[key], e.continuum[key], rtol=1e-11)
    assert t.continuum["LH2O"] == e.continuum["LH2O"]
    for key in ("radii", "w0", "qext", "gt"):
        np.testing.assert_allclose(t.particles[0][key], e.particles[0][key], rtol=1e-11)
    np.testing.assert_allclose(t.photons_sol, e.photons_sol, rtol=1e-11)
    # the padding value of the regridding is the reference's log10(sqrt(tiny)) (src/clima_const.f90:21):
    # the first bin is only partly covered by the CIA data, so its average shows the pad
    assert D.LOG10TINY == pytest.approx(-153.8263277842944, rel=1e-15)
    cia = [x for x in t.xsections if x["xs_type"] == D.XS_CIA][0]["data"]
    assert -130.0 < cia[0, 0] < -60.0


def test_loaded_tables_drive_the_oracle(datadir, O):
    # what the loader returns is a complete table set: the CPU oracle runs on it
    from clima_amd import data_loader as D
    root, tb, _ = datadir
    t = D.load_tables(os.path.join(root, "settings.yaml"), os.path.join(root, "star.txt"), root)
    o = O.OracleRadtran(t, 20, 2, 0.2)
    isr, olr = o.TOA_fluxes(*S.modern_earth_column(20).args())
    assert np.isfinite(isr) and np.isfinite(olr) and olr > 0.0 and isr > 0.0


def test_reference_error_texts(datadir, tmp_path):
    from clima_amd import data_loader as D
    from clima_amd.radtran import ClimaException
    root, tb, _ = datadir
    sp = list(tb.species_names)

    def sop(opacities, kmethod="RandomOverlapResortRebin"):
        return D.SettingsOpacity({"k-method": kmethod, "opacities": opacities})

    with pytest.raises(ClimaException, match='k-method "Foo" in "settings" is not an option.'):
        sop({"k-distributions": True}, "Foo")
    with pytest.raises(ClimaException, match='"H2O" is a duplicate in k-distributions'):
        sop({"k-distributions": ["H2O", "CO2", "H2O"]})
    with pytest.raises(ClimaException, match="You must specify at least one k-distribution"):
        D.create_optical_properties(root, sp, [], sop({"CIA": True}))
    with pytest.raises(ClimaException, match='Species "Xe" in optical property "k-distributions" is not in the list'):
        D.create_optical_properties(root, sp, [], sop({"k-distributions": ["Xe"]}))
    import shutil
    shutil.copy(os.path.join(root, "CIA", "N2-N2.h5"), os.path.join(root, "CIA", "H2O-H2O.h5"))
    with pytest.raises(ClimaException, match="double count opacity"):
        D.create_optical_properties(root, sp + ["H2"], [], sop({"k-distributions": ["H2O"], "CIA": ["H2O-H2O"],
                                                               "water-continuum": "MT_CKD"}))
    with pytest.raises(ClimaException, match='Could not parse CIA species pair "N2-Xe" into two known species.'):
        D.create_optical_properties(root, sp, [], sop({"k-distributions": ["H2O"], "CIA": ["N2-Xe"]}))
    with pytest.raises(ClimaException, match='Continuum "CKD9" is not avaliable.'):
        D.create_optical_properties(root, sp, [], sop({"k-distributions": ["H2O"], "water-continuum": "CKD9"}))
    with pytest.raises(ClimaException, match='Failed to read'):
        D.read_ktable(os.path.join(root, "settings.yaml"), 0)
    with pytest.raises(ClimaException, match="does not exist."):
        D.read_stellar_flux(str(tmp_path / "nostar.txt"), tb.sol_wavl)
    # "on" picks up exactly what the directory holds; H2O pairs are left out next to the continuum
    t = D.create_optical_properties(root, sp, ["HCaer1"], sop({"k-distributions": True, "CIA": True,
                                                                "rayleigh": True, "photolysis-xs": True,
                                                                "water-continuum": "MT_CKD"}))
    assert len(t.ktables) == 5 and sum(x["xs_type"] == D.XS_CIA for x in t.xsections) == 6
    assert sum(x["xs_type"] == D.XS_RAYLEIGH for x in t.xsections) == 5
