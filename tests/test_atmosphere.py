"""Caller-side glue (clima_amd/atmosphere.py): atmosphere-file reader, unpacking onto the
vertical grid, radiative-grid doubling -- SURVEY.md 8(f) next #2."""
import os

import numpy as np
import pytest

from clima_amd import synthetic as S

REF_TEMPLATE = "/root/reference/templates/ModernEarth/atmosphere.txt"


def _write_atmosphere(path, drop=None, rename=None):
    d = np.load(os.path.join(os.path.dirname(S.__file__), "data", "modern_earth_atmosphere.npz"))
    names = [str(s) for s in d["species"]]
    labels = ["alt", "press", "temp"] + names
    cols = [d["alt_km"], d["press_bar"], d["temp_K"]] + [d["mix"][:, i] for i in range(len(names))]
    if drop:
        keep = [i for i, l in enumerate(labels) if l != drop]
        labels, cols = [labels[i] for i in keep], [cols[i] for i in keep]
    if rename:
        labels = [rename.get(l, l) for l in labels]
    with open(path, "w") as f:
        f.write("".join("%-28s" % l for l in labels) + "\n")
        for r in range(len(cols[0])):
            f.write("".join("%-28s" % ("%.17e" % c[r]) for c in cols) + "\n")
    return labels


def test_file_column_equals_fixture_column(tmp_path):
    from clima_amd.atmosphere import AtmosphereFile, column_from_atmosphere
    p = str(tmp_path / "atmosphere.txt")
    labels = _write_atmosphere(p)
    atm = AtmosphereFile(p)
    assert atm.labels == labels and atm.nz == 200 and atm.columns.shape == (len(labels), 200)
    for nz in (200, 50):
        col = column_from_atmosphere(atm, S.MODERN_EARTH_SPECIES, nz=nz, n_particles=1)
        ref = S.modern_earth_column(nz)
        for k in ("T", "P", "dz", "densities", "pdensities", "radii"):
            np.testing.assert_array_equal(col[k], ref[k])
        assert col["T_surface"] == ref["T_surface"]


@pytest.mark.skipif(not os.path.exists(REF_TEMPLATE), reason="reference templates not present")
def test_reference_template_parses_to_the_fixture():
    from clima_amd.atmosphere import column_from_atmosphere
    col = column_from_atmosphere(REF_TEMPLATE, S.MODERN_EARTH_SPECIES, n_particles=1)
    ref = S.modern_earth_column(200)
    for k in ("T", "P", "dz", "densities"):
        np.testing.assert_allclose(col[k], ref[k], rtol=1e-12)


def test_radiative_grid_doubling():
    from clima_amd.atmosphere import copy_atm_to_radiative_grid
    col = S.modern_earth_column(7)
    r = copy_atm_to_radiative_grid(col)
    nz = 7
    assert len(r["T"]) == 2 * nz + 2 and r["densities"].shape == (2 * nz + 2, col["densities"].shape[1])
    for i in range(nz):
        for k in ("T", "P"):
            assert r[k][2 * i] == r[k][2 * i + 1] == col[k][i]
        assert r["dz"][2 * i] == r["dz"][2 * i + 1] == 0.5 * col["dz"][i]
        np.testing.assert_array_equal(r["densities"][2 * i], col["densities"][i])
    for g in (2 * nz, 2 * nz + 1):   # ghost layers copy the top RT layer
        assert r["T"][g] == col["T"][-1] and r["dz"][g] == 0.5 * col["dz"][-1]
        np.testing.assert_array_equal(r["radii"][g], col["radii"][-1])
    assert abs(r["dz"][: 2 * nz].sum() - col["dz"].sum()) < 1e-6
    same = copy_atm_to_radiative_grid(col, double_radiative_grid=False)
    np.testing.assert_array_equal(same["T"], col["T"])


def test_reference_error_texts(tmp_path):
    from clima_amd.atmosphere import AtmosphereFile, unpack_atmospherefile, vertical_grid
    from clima_amd.radtran import ClimaException
    z, _ = vertical_grid(0.0, 1.0e7, 10)
    with pytest.raises(ClimaException, match="Can not open file"):
        AtmosphereFile(str(tmp_path / "absent.txt"))
    p = str(tmp_path / "a.txt")
    _write_atmosphere(p, drop="CH4")
    with pytest.raises(ClimaException, match='Species "CH4" was not found in'):
        unpack_atmospherefile(AtmosphereFile(p), list(S.MODERN_EARTH_SPECIES), z)
    _write_atmosphere(p, rename={"temp": "tmp"})
    with pytest.raises(ClimaException, match='"temp" was not found in input file'):
        unpack_atmospherefile(AtmosphereFile(p), list(S.MODERN_EARTH_SPECIES), z)
    _write_atmosphere(p, rename={"alt": "height"})
    with pytest.raises(ClimaException, match='"alt" was not found in input file'):
        unpack_atmospherefile(AtmosphereFile(p), list(S.MODERN_EARTH_SPECIES), z)
    _write_atmosphere(p)
    with pytest.raises(ClimaException, match="mixing ratios do not sum to close to 1"):
        unpack_atmospherefile(AtmosphereFile(p), ["N2", "O2"][:1], z)
    with open(p, "a") as f:
        f.write("1.0 2.0\n")
    with pytest.raises(ClimaException, match="Problem reading in initial atmosphere"):
        AtmosphereFile(p)
