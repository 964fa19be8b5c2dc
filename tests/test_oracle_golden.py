"""The oracle's two-stream solver against the REFERENCE's own (compiled) two-stream.

tests/golden/twostream_golden.npz holds outputs of the unmodified reference
src/radtran/clima_radtran_twostream.f90 (see tests/golden/make_twostream_golden.py).
This is the pin that ties the oracle -- and through it the HIP path -- to the reference.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "twostream_golden.npz")
# The restatement keeps the reference's operation order; gcc -ffp-contract=off vs flang -O2
# leaves only libm/ulp-level differences.
RTOL = 1e-13


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(np.abs(b), 1e-300))


def test_golden_two_stream_ir(O):
    d = np.load(GOLD)
    for n in range(int(d["ncases"][0])):
        k = "c%02d_" % n
        em, hs, tmin = d[k + "ir_par"]
        fup, fdn = O.two_stream_ir(d[k + "tau"], d[k + "w0"], d[k + "g"], em, bool(hs), tmin, d[k + "bplanck"])
        assert _rel(fup, d[k + "ir_fup"]) <= RTOL, n
        assert _rel(fdn[1:], d[k + "ir_fdn"][1:]) <= RTOL, n
        assert fdn[0] == 0.0 == d[k + "ir_fdn"][0]


def test_golden_two_stream_ir_changed_profiles(O):
    """The response cases: 31 of the smooth / paired columns with a few Planck values changed each (279 profiles)."""
    d = np.load(GOLD)
    nresp = int(d["nresp"][0])
    assert nresp >= 25
    for m in range(nresp):
        r = "r%02d_" % m
        k = "c%02d_" % int(d[r + "base"][0])
        em, hs, tmin = d[k + "ir_par"]
        for j in range(int(d[r + "ncol"][0])):
            bp = d[k + "bplanck"].copy()
            bp[d[r + "c%d_k" % j]] = d[r + "c%d_b" % j]
            fup, fdn = O.two_stream_ir(d[k + "tau"], d[k + "w0"], d[k + "g"], em, bool(hs), tmin, bp)
            assert _rel(fup, d[r + "c%d_fup" % j]) <= RTOL, (m, j)
            assert _rel(fdn[1:], d[r + "c%d_fdn" % j][1:]) <= RTOL, (m, j)


def test_golden_two_stream_solar(O):
    d = np.load(GOLD)
    for n in range(int(d["ncases"][0])):
        k = "c%02d_" % n
        u0, rs = d[k + "sol_par"]
        am, sr, fup, fdn = O.two_stream_solar(d[k + "tau"], d[k + "w0"], d[k + "g"], u0, rs)
        assert _rel(am, d[k + "sol_amean"]) <= RTOL, n
        assert _rel(fup, d[k + "sol_fup"]) <= RTOL, n
        assert _rel(fdn, d[k + "sol_fdn"]) <= RTOL, n
        assert _rel(sr, d[k + "sol_sr"][0]) <= RTOL, n


def test_live_reference_if_built(O):
    """In the build container the compiled reference itself is available: random columns."""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(3)
    for nz in (1, 4, 33, 200):
        for _ in range(20):
            tau = 10 ** rng.uniform(-7, 2, nz)
            w0 = rng.uniform(0, 0.99999, nz)
            g = rng.uniform(0, 0.8, nz)
            bp = rng.uniform(1e-12, 1e-9, nz + 1)
            a = O.two_stream_ir(tau, w0, g, 0.9, True, 1e-6, bp)
            b = O.ref_two_stream_ir(tau, w0, g, 0.9, True, 1e-6, bp)
            assert _rel(a[0], b[0]) <= RTOL
            u0 = rng.uniform(0.05, 1)
            a = O.two_stream_solar(tau, w0, g, u0, 0.2)
            b = O.ref_two_stream_solar(tau, w0, g, u0, 0.2)
            for x, y in zip((a[0], a[2], a[3]), (b[0], b[2], b[3])):
                assert _rel(x, y) <= RTOL
