"""radtran_radiate_ir_batch, response form (clima_amd/csrc/ir_green.inc): columns that differ from the batch's majority
profile in a few temperatures are F(base) + unit responses x Planck differences.  Held to the oracle's full solves
(two_stream_ir, src/radtran/clima_radtran_twostream.f90:156-295, one per column as
src/adiabat/clima_adiabat_solve.f90:798-812 issues them) and to the library's general batch kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_LEVEL = 2.0e-11     # of the row's largest value, as in test_gpu_parity


def _scaled(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _jacobian_batch(col, nz, ncol, rng, extra=True):
    T = np.repeat(np.asarray(col["T"], dtype=float)[:, None], ncol, axis=1)
    Ts = np.full(ncol, float(col["T_surface"]))
    for c in range(ncol):
        k = c % (nz + 1)
        dT = 1.0e-2 * (1.0 + rng.random()) * (Ts[c] if k == 0 else T[k - 1, c])
        if k == 0:
            Ts[c] += dT
        else:
            T[k - 1, c] += dT
    if extra and ncol >= 8:
        T[:, 1] = T[:, 1] * (1.0 + 0.01 * rng.random(nz))      # a dense column: every layer moved (general kernel)
        Ts[2] = float(col["T_surface"]); T[:, 2] = col["T"]      # the base itself
        for j in (0, nz // 2, nz - 1):                          # three deviations in one column, top and bottom layers among them
            T[j, 3] += 0.7
        T[nz - 1, 4] += 0.3; Ts[4] += 0.4                       # top layer and surface
        T[0, 5] -= 0.5; T[1, 5] += 0.5                          # neighbours
    return Ts, T


@pytest.mark.parametrize("nz,ncol,hard", [(4, 9, True), (5, 12, False), (30, 70, True), (50, 60, False), (64, 20, True),
                                          (102, 110, True), (130, 30, False), (202, 40, True), (402, 24, True)])
def test_response_form_against_the_oracle_and_the_general_kernel(O, small_tables, nz, ncol, hard):
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    from test_gpu_parity import _pair, _compare
    col = S.modern_earth_column(nz)
    r, o = _pair(O, small_tables, nz, 2, 0.3)
    r.has_hard_surface = hard
    o.set_scalars(has_hard_surface=hard)
    _compare(r, o, col)
    rng = np.random.default_rng(nz * 1000 + ncol)
    Ts, T = _jacobian_batch(col, nz, ncol, rng)
    r.ir_green = 0
    gen = r.radiate_ir_batch(Ts, T)
    assert r.ir_green_batches == 0
    r.ir_green = 2
    got = r.radiate_ir_batch(Ts, T)
    assert r.ir_green_batches == 1
    for a, b in zip(got, gen):
        for c in range(ncol):
            assert _scaled(a[:, c], b[:, c]) <= 1.0e-11, (c, _scaled(a[:, c], b[:, c]))
    for c in range(min(ncol, 14)):
        w = S.Column(col)
        w["T"] = T[:, c].copy()
        w["T_surface"] = Ts[c]
        o.radiate(*w.args(), compute_solar=False, compute_opacity=False)
        assert _scaled(got[0][:, c], o.wrk_ir.fup_n) <= TOL_LEVEL
        assert _scaled(got[1][:, c], o.wrk_ir.fdn_n) <= TOL_LEVEL
        assert _scaled(got[2][:, c], o.f_total) <= TOL_LEVEL


def test_response_form_jacobian_entries(O, small_tables):
    """What the Jacobian is made of: (F(T + dT) - F(T)) / dT per level.  The response form computes the difference
    itself instead of subtracting two solves, so its finite differences must agree with the oracle's to the oracle's own
    cancellation error (~1e-12 |F| / dT after the sums over bins and g-points), here with perturbations of 1e-4 relative."""
    from clima_amd import synthetic as S
    from test_gpu_parity import _pair, _compare
    nz = 60
    col = S.modern_earth_column(nz)
    r, o = _pair(O, small_tables, nz, 2, 0.3)
    _compare(r, o, col)
    ncol = nz + 2
    T = np.repeat(np.asarray(col["T"], dtype=float)[:, None], ncol, axis=1)
    Ts = np.full(ncol, float(col["T_surface"]))
    dT = np.zeros(ncol)
    for c in range(1, ncol):
        k = c - 1
        if k == 0:
            dT[c] = 1e-4 * Ts[c]; Ts[c] += dT[c]
        else:
            dT[c] = 1e-4 * T[k - 1, c]; T[k - 1, c] += dT[c]
    r.ir_green = 2
    fup, fdn, ftot = r.radiate_ir_batch(Ts, T)
    assert r.ir_green_batches == 1
    o.radiate(*col.args(), compute_solar=False, compute_opacity=False)
    base = np.array(o.f_total)
    for c in (1, 2, nz // 2, nz + 1):
        w = S.Column(col)
        w["T"] = T[:, c].copy()
        w["T_surface"] = Ts[c]
        o.radiate(*w.args(), compute_solar=False, compute_opacity=False)
        want = (np.array(o.f_total) - base) / dT[c]
        have = (ftot[:, c] - ftot[:, 0]) / dT[c]
        scale = np.max(np.abs(want))
        assert np.max(np.abs(have - want)) <= 1e-7 * scale + 3e-12 * np.max(np.abs(base)) / dT[c]


def test_response_form_is_chosen_by_itself_for_a_jacobian_sized_batch():
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    nz = 210                                         # (a small spectrum or a short grid: the general kernel is as fast)
    col = S.modern_earth_column(nz)
    r = Radtran(S.modern_earth_tables(nw=400), nz, 2, 0.3)
    r.radiate(*col.args())
    assert r.ir_green == 1
    rng = np.random.default_rng(3)
    Ts, T = _jacobian_batch(col, nz, 100, rng)
    a = r.radiate_ir_batch(Ts, T)
    assert r.ir_green_batches == 1
    Ts2, T2 = _jacobian_batch(col, nz, 20, rng)      # too few columns: the general kernel
    r.radiate_ir_batch(Ts2, T2)
    assert r.ir_green_batches == 1
    Ts4, T4 = Ts.copy(), T.copy()                    # eight changes per column: the accumulation would cost more than it saves
    for c in range(T4.shape[1]):
        T4[rng.integers(0, nz, 7), c] += 0.25
    r.radiate_ir_batch(Ts4, T4)
    assert r.ir_green_batches == 1
    T3 = T * (1.0 + 0.01 * rng.random(T.shape))       # nothing in common: the general kernel
    r.radiate_ir_batch(Ts, T3)
    assert r.ir_green_batches == 1
    r.ir_green = 0
    b = r.radiate_ir_batch(Ts, T)
    for x, y in zip(a, b):
        assert _scaled(x, y) <= 1e-11


@pytest.mark.parametrize("ng,nz", [(4, 50), (6, 60), (12, 70), (16, 130), (32, 40)])
def test_response_form_at_other_g_point_counts(O, ng, nz):
    from clima_amd import synthetic as S
    from test_gpu_parity import _pair
    tb = S.modern_earth_tables(nw=12, ng=ng, seed=40 + ng)
    col = S.modern_earth_column(nz)
    r, o = _pair(O, tb, nz, 2, 0.25)
    r.radiate(*col.args())
    o.radiate(*col.args())
    rng = np.random.default_rng(ng)
    Ts, T = _jacobian_batch(col, nz, 12, rng)
    r.ir_green = 2
    fup, fdn, ftot = r.radiate_ir_batch(Ts, T)
    assert r.ir_green_batches == 1
    for c in range(12):
        w = S.Column(col)
        w["T"] = T[:, c].copy()
        w["T_surface"] = Ts[c]
        o.radiate(*w.args(), compute_solar=False, compute_opacity=False)
        assert _scaled(fup[:, c], o.wrk_ir.fup_n) <= TOL_LEVEL
        assert _scaled(fdn[:, c], o.wrk_ir.fdn_n) <= TOL_LEVEL
        assert _scaled(ftot[:, c], o.f_total) <= TOL_LEVEL


def test_response_form_with_extreme_optical_depths(O, small_tables):
    """Layers that transmit nothing (exp(-lambda tau) underflows to 0: phi = 0 exactly) under layers that are nearly
    transparent: the running products are held away from 0 and the ratios stay ratios.  Custom optical properties
    (clima_radtran.f90:494-506) put the thick layers there."""
    from clima_amd import synthetic as S
    from test_gpu_parity import _pair
    nz = 40
    col = S.modern_earth_column(nz)
    r, o = _pair(O, small_tables, nz, 2, 0.3)
    wv = np.geomspace(150.0, 4.0e5, 7)                            # nm
    P = np.geomspace(1.2e6, 0.5, 6)                               # dynes/cm^2, decreasing
    dtau_dz = np.repeat((10.0 ** np.linspace(-1.5, -13.0, 6))[:, None], 7, axis=1)   # 1/cm: thousands per layer at the ground
    w0 = np.full((6, 7), 0.3)
    g0 = np.full((6, 7), 0.2)
    for x in (r, o):
        x.set_custom_optical_properties(wv, P, dtau_dz, w0, g0)
    r.radiate(*col.args())
    o.radiate(*col.args())
    assert np.max(r.opr()[0]) > 2000.0                # exp(-lambda tau) = 0 there
    rng = np.random.default_rng(5)
    Ts, T = _jacobian_batch(col, nz, nz + 1, rng, extra=False)
    r.ir_green = 0
    gen = r.radiate_ir_batch(Ts, T)
    r.ir_green = 2
    got = r.radiate_ir_batch(Ts, T)
    assert r.ir_green_batches == 1
    for a, b in zip(got, gen):
        assert np.all(np.isfinite(a))
        for c in range(nz + 1):
            assert _scaled(a[:, c], b[:, c]) <= 1.0e-11, c
    for c in (0, 1, nz // 2, nz):
        w = S.Column(col)
        w["T"] = T[:, c].copy()
        w["T_surface"] = Ts[c]
        o.radiate(*w.args(), compute_solar=False, compute_opacity=False)
        assert _scaled(got[0][:, c], o.wrk_ir.fup_n) <= TOL_LEVEL
        assert _scaled(got[2][:, c], o.f_total) <= TOL_LEVEL


def test_a_profile_that_is_not_a_number_stays_with_the_general_kernel(small_tables):
    """NaN in the shared profile: every column would "deviate" at that level and inherit it from the base solve; the
    general kernel confines it to the columns that carry it."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    nz = 30
    col = S.modern_earth_column(nz)
    r = Radtran(small_tables, nz, 2, 0.3)
    r.radiate(*col.args())
    rng = np.random.default_rng(2)
    Ts, T = _jacobian_batch(col, nz, 12, rng, extra=False)
    T[5, :] = np.nan
    r.ir_green = 2
    r.radiate_ir_batch(Ts, T)
    assert r.ir_green_batches == 0
    T[5, :] = col["T"][5]
    T[5, 3] = np.nan                                   # one column only: a deviation like any other
    a = r.radiate_ir_batch(Ts, T)
    assert r.ir_green_batches == 1
    assert np.all(np.isnan(a[0][:, 3])) or np.any(np.isnan(a[0][:, 3]))
    assert np.all(np.isfinite(np.delete(a[0], 3, axis=1)))


def test_response_form_with_steps_far_below_what_two_solves_can_resolve(O, small_tables):
    """A Jacobian's finite-difference step can be 1e-8 T and smaller.  The Planck difference of such a step is formed
    without subtracting two nearly equal Planck values (k_green_db), and the response form never subtracts two solves: its
    difference quotients must stay on the derivative -- taken here from the oracle's central differences at a comfortable
    step -- down to where the returned rows themselves (F(T + dT), not the change) stop resolving it."""
    from clima_amd import synthetic as S
    from test_gpu_parity import _pair, _compare
    nz = 40
    col = S.modern_earth_column(nz)
    r, o = _pair(O, small_tables, nz, 2, 0.3)
    _compare(r, o, col)
    levels = [0, 3, nz // 2, nz]                       # T index (nz = the surface)
    rel_steps = [1e-6, 1e-9, 1e-12]
    ncol = 1 + len(levels) * len(rel_steps)
    T = np.repeat(np.asarray(col["T"], dtype=float)[:, None], ncol, axis=1)
    Ts = np.full(ncol, float(col["T_surface"]))
    dT = np.zeros(ncol)
    c = 1
    for j in levels:
        for rs in rel_steps:
            base = Ts[c] if j == nz else T[j, c]
            dT[c] = rs * base
            if j == nz:
                Ts[c] = base + dT[c]
                dT[c] = Ts[c] - base                    # the step as the floating-point numbers carry it
            else:
                T[j, c] = base + dT[c]
                dT[c] = T[j, c] - base
            c += 1
    r.ir_green = 2
    fup, fdn, ftot = r.radiate_ir_batch(Ts, T)
    assert r.ir_green_batches == 1
    c = 1
    for j in levels:
        # central difference of the oracle at a relative step of 1e-5: truncation ~1e-8 (x^2 step^2 / 6, x = h nu / k T
        # up to ~20), cancellation ~1e-8
        d = 1e-5 * (float(col["T_surface"]) if j == nz else float(col["T"][j]))
        rows = []
        for sgn in (+1.0, -1.0):
            w = S.Column(col)
            w["T"] = np.asarray(col["T"], dtype=float).copy()
            if j == nz:
                w["T_surface"] = float(col["T_surface"]) + sgn * d
            else:
                w["T"][j] += sgn * d
            o.radiate(*w.args(), compute_solar=False, compute_opacity=False)
            rows.append(np.array(o.f_total))
        want = (rows[0] - rows[1]) / (2.0 * d)
        for rs in rel_steps:
            have = (ftot[:, c] - ftot[:, 0]) / dT[c]
            # the one-sided quotient's own truncation (~x^2 / 2 times the relative step) and the resolution of the RESULT:
            # the batch returns F(T + dT), not the change, so the caller's subtraction keeps ~1e-16 F / dF
            tol = 2e-6 + 500.0 * rs + 3e-15 / rs
            assert np.max(np.abs(have - want)) <= tol * np.max(np.abs(want)), (j, rs, np.max(np.abs(have - want)) / np.max(np.abs(want)))
            c += 1


@pytest.mark.parametrize("nz,ncol,ng", [(202, 230, 8), (130, 150, 8), (70, 90, 6), (40, 50, 3)])
def test_far_accumulation_on_the_matrix_cores_against_the_vector_kernel(nz, ncol, ng):
    """k_green_accum_far_mfma (v_mfma_f64_16x16x4_f64: one wave = 64 consecutive deviations of a block pair's below-form
    prefix or above-form suffix, four q per K step) against k_green_accum_far (vector FMAs, fixed groups of 64
    deviations, both forms per group) on batches with several chunks per block, a last chunk that is not full, splits
    whose q count is not a multiple of four (6 and 3 g-points) and an odd number of level blocks (40 and 70 layers: the last
    pair has one block).  The two sum the same terms in different orders: rounding only."""
    import ctypes as C
    from clima_amd import synthetic as S
    from clima_amd.lib import load
    from clima_amd.radtran import Radtran
    L = load()
    tb = S.modern_earth_tables(nw=24, ng=ng, seed=7 + ng) if ng != 8 else S.modern_earth_tables(nw=40, seed=11)
    col = S.modern_earth_column(nz)
    r = Radtran(tb, nz, 2, 0.2)
    r.radiate(*col.args())
    rng = np.random.default_rng(nz + ncol)
    Ts, T = _jacobian_batch(col, nz, ncol, rng)
    r.ir_green = 2
    res = {}
    try:
        for form in (1, 0):
            L.clima_test_green_far_form_set(C.byref(C.c_int(form)))
            res[form] = r.radiate_ir_batch(Ts, T)
    finally:
        L.clima_test_green_far_form_set(C.byref(C.c_int(0)))
    assert r.ir_green_batches == 2
    for a, b in zip(res[0], res[1]):
        for c in range(ncol):
            assert _scaled(a[:, c], b[:, c]) <= 2.0e-13, (c, _scaled(a[:, c], b[:, c]))


def test_batch_results_in_page_locked_caller_arrays(small_tables):
    """radtran_batch_pin_results_set: from the second call with the same three result arrays on they are page-locked and
    filled by the device directly; by default the results come through the handle's pinned block in pieces.  Same
    values either way, bit for bit (the same device arrays are copied), for the response form and the general kernel."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    nz, ncol = 70, 90
    col = S.modern_earth_column(nz)
    r = Radtran(small_tables, nz, 2, 0.25)
    r.radiate(*col.args())
    Ts, T = _jacobian_batch(col, nz, ncol, np.random.default_rng(5))
    for mode in (2, 0):
        r.ir_green = mode
        ref = r.radiate_ir_batch(Ts, T)
        out = [np.full((nz + 1, ncol), np.nan, order="F") for _ in range(3)]
        for rep in range(3):
            for o in out:
                o[...] = np.nan
            got = r.radiate_ir_batch(Ts, T, out=out, pin=True)
            for a, b in zip(got, ref):
                assert np.array_equal(a, b), (mode, rep)
        r.spectra_release()
        again = r.radiate_ir_batch(Ts, T, out=out)        # unpinned again: the same arrays, now pageable
        for a, b in zip(again, ref):
            assert np.array_equal(a, b)


def test_thousands_of_deviations_in_few_bin_splits():
    """420 columns x 8 changed levels on a 130-layer grid: 3 360 deviations, so many chunks per level block that the far
    accumulation's waves fill the machine with fewer than 8 bin splits -- its chunks then go round the XCDs instead of
    its splits (the other mapping idled 6 of 8 XCDs: 5.96 ms where 1.2 are due).  Matrix form against the vector form and
    both against the general kernel."""
    import ctypes as C
    from clima_amd import synthetic as S
    from clima_amd.lib import load
    from clima_amd.radtran import Radtran
    L = load()
    nz, ncol, per = 130, 420, 8
    tb = S.modern_earth_tables(nw=40, seed=13)
    col = S.modern_earth_column(nz)
    r = Radtran(tb, nz, 2, 0.2)
    r.radiate(*col.args())
    rng = np.random.default_rng(99)
    T = np.repeat(np.asarray(col["T"], dtype=float)[:, None], ncol, axis=1)
    Ts = np.full(ncol, float(col["T_surface"]))
    for c in range(ncol):
        for j in rng.choice(nz + 1, per, replace=False):
            if j == nz:
                Ts[c] += rng.uniform(-3, 3)
            else:
                T[j, c] += rng.uniform(-3, 3)
    r.ir_green = 0
    gen = r.radiate_ir_batch(Ts, T)
    r.ir_green = 2
    res = {}
    try:
        for form in (1, 0):
            L.clima_test_green_far_form_set(C.byref(C.c_int(form)))
            res[form] = r.radiate_ir_batch(Ts, T)
    finally:
        L.clima_test_green_far_form_set(C.byref(C.c_int(0)))
    assert r.ir_green_batches == 2
    for a, b, g in zip(res[0], res[1], gen):
        for c in range(ncol):
            assert _scaled(a[:, c], b[:, c]) <= 2.0e-13, (c, _scaled(a[:, c], b[:, c]))
            assert _scaled(a[:, c], g[:, c]) <= 1.0e-11, (c, _scaled(a[:, c], g[:, c]))
