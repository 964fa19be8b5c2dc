"""Distinct handles are independent ("Threading": the reference's object is not thread-safe but two
objects are -- it is built with OpenMP, /root/reference/src/CMakeLists.txt:78-86, and its Python package pins
OMP_NUM_THREADS for its own calls only, clima/__init__.py:1-2).  Here a handle owns its stream, its device buffers
and its hand-off flags, so two handles may be driven from two host threads, or interleaved from one, and their
fused grids may share the machine: a two-stream block of one handle then waits (bounded spin on its opacity tiles'
flags) while blocks of the OTHER handle hold wave slots.  Every result must be bitwise what the handle computes
alone, and no wait may expire (radtran_fused_fallbacks_get)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cols(nz, n):
    from clima_amd import synthetic as S
    return S.perturbed_columns(n, nz=nz, seed=11)


def _rows(r):
    return np.concatenate([np.asarray(r.wrk_ir.fup_n), np.asarray(r.wrk_ir.fdn_n), np.asarray(r.wrk_sol.fup_n),
                           np.asarray(r.wrk_sol.fdn_n), np.asarray(r.f_total)])


def test_two_handles_on_two_host_threads():
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    nz, ncalls = 200, 200
    tables = S.modern_earth_tables()                  # config 2's tables: the fused grid fills the machine by itself
    cols = _cols(nz, 4)
    rads = [Radtran(tables, nz, 8, 0.15), Radtran(tables, nz, 8, 0.15)]
    assert all(r.fused for r in rads)
    # serial results of each handle on its two columns
    want = []
    for h, r in enumerate(rads):
        w = []
        for c in (cols[2 * h], cols[2 * h + 1]):
            toa = r.TOA_fluxes(*c.args())
            w.append((toa, _rows(r)))
        want.append(w)
    base = [r.fused_fallbacks for r in rads]
    errors = []

    def work(h):
        try:
            r = rads[h]
            for i in range(ncalls):
                c = cols[2 * h + (i & 1)]
                toa = r.TOA_fluxes(*c.args())
                wt, wr = want[h][i & 1]
                if toa != wt or not np.array_equal(_rows(r), wr):
                    errors.append("handle %d call %d differs from its serial result" % (h, i))
                    return
        except Exception as e:   # noqa: BLE001
            errors.append("handle %d: %r" % (h, e))
    ts = [threading.Thread(target=work, args=(h,)) for h in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert [r.fused_fallbacks for r in rads] == base, "a bounded hand-off wait expired with two grids on the device"


def test_two_handles_interleaved_from_one_thread():
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    nz = 200
    tables = S.modern_earth_tables()
    cols = _cols(nz, 2)
    rads = [Radtran(tables, nz, 8, 0.15), Radtran(tables, nz, 8, 0.15)]
    want = []
    for r, c in zip(rads, cols):
        r.upload_column(*c.args())
        r.radiate_resident()
        r.synchronize()
        want.append(_rows(r))
    base = [r.fused_fallbacks for r in rads]
    for rep in range(50):
        # four calls of each handle enqueued alternately on the two streams, nothing synchronised in between
        for _ in range(4):
            for r in rads:
                r.radiate_resident()
        for r in rads:
            r.synchronize()
        for h, r in enumerate(rads):
            np.testing.assert_array_equal(_rows(r), want[h], err_msg="handle %d, repetition %d" % (h, rep))
    assert [r.fused_fallbacks for r in rads] == base


def test_all_spectra_in_one_go_equal_the_seven_getters():
    """radtran_spectra_get_all (the Fortran module's default way to fill rad%wrk_*%fup_a ... after a call: caller arrays
    page-locked once, asynchronous copies, one synchronise) against the reference-named getters, twice into the same
    arrays and once IR-only."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    nz = 60
    tables = S.modern_earth_tables(nw=80)
    r = Radtran(tables, nz, 4, 0.2)
    cols = _cols(nz, 2)
    out = None
    for c in cols:
        r.TOA_fluxes(*c.args())
        out = r.spectra_all(out=out)
        for ch, w in (("ir", r.wrk_ir), ("sol", r.wrk_sol)):
            np.testing.assert_array_equal(out[ch + "_fup_a"], np.asarray(w.fup_a))
            np.testing.assert_array_equal(out[ch + "_fdn_a"], np.asarray(w.fdn_a))
            np.testing.assert_array_equal(out[ch + "_tau_band"], np.asarray(w.tau_band))
        np.testing.assert_array_equal(out["sol_amean"], np.asarray(r.wrk_sol.amean))
    keep = out["sol_fup_a"].copy()
    r.radiate(*cols[0].args(), compute_solar=False)
    out = r.spectra_all(do_solar=False, out=out)
    np.testing.assert_array_equal(out["ir_fup_a"], np.asarray(r.wrk_ir.fup_a))
    np.testing.assert_array_equal(out["sol_fup_a"], keep)          # not touched by an IR-only fetch
    r.spectra_release()
