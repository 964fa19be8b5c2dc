"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Floating-point path: tolerances are stated here.

north_star tolerance: OLR within 1e-4 relative of the reference.  What is enforced:
  * OLR / ISR ............................ 1e-9 relative
  * level fluxes fup_n, fdn_n, f_total .... 1e-9 of the channel's profile maximum (up and down together)
  * per-bin spectra fup_a, fdn_a, amean ... 1e-8 of the array maximum (fup_a, fdn_a together)
  * opr tau, w0, g, tau_band .............. 1e-11 relative (k-table exp() argument rounding
    bounds this at ~3e-14; measured 2.8e-14)
Differences come from device exp/log10 (<=1 ulp), FMA contraction, reciprocal-multiply in
the chunked Thomas sweeps, and the summation order of the zenith/g-point weights.

Three DELIBERATE departures from the reference's arithmetic sit inside these tolerances:
  * the 64 sums of a random-overlap mixing step carry their pair index in mantissa bits 3-8 and that key IS the value
    that is rebinned (<= 2^-44 relative; clima_amd/csrc/kernels.hip KEY_IDX_MASK);
  * the fused grid's two-stream part forms w0 = min(0.99999, scat * (1/tau)) with the correctly rounded reciprocal
    (<= 1 ulp from the quotient of clima_radtran_types.f90:869-875; the stored w0 array is the quotient itself);
  * round 4: rows of a mixing step that stand alone above everything before them are rebinned as sum_j w_j key(k, j)
    instead of the difference quotient of the running integral (1e-15 relative, the more accurate of the two) -- which
    rows do is decided per 64-lane wave, so a lane's last bits depend on its wave-mates
    (test_mixing_step_with_every_kind_of_wave below forces every case).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL_TOA = 1e-9
TOL_LEVEL = 1e-9
TOL_SPEC = 1e-8
RTOL_OPR = 1e-11


def _rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def _scaled(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def _pair(O, tables, nz, nzen, albedo, **scalars):
    from clima_amd.radtran import Radtran
    r = Radtran(tables, nz, nzen, albedo)
    o = O.OracleRadtran(tables, nz, nzen, albedo)
    for k, v in scalars.items():
        setattr(r, k, v)
    if scalars:
        o.set_scalars(**scalars)
    return r, o


def _compare(r, o, col, flux_tol_scale=1.0, **kw):
    """HIP against the oracle.  With 8 g-points a call of few (bin, layer) items runs the group-of-lanes
    opacity kernel (k_opacity_coop<8>) and one launch per kernel, a larger one the lane-per-item kernel
    inside the fused grid: a small test case is therefore compared TWICE, once in each form."""
    out = _compare_once(r, o, col, flux_tol_scale, **kw)
    items = r.coop_items
    if r.ngauss == 8 and items > 0 and r.nw * r.nz <= items and kw.get("compute_opacity", True):
        r.coop_items = 0
        try:
            _compare_once(r, o, col, flux_tol_scale, **kw)
        finally:
            r.coop_items = items
    return out


def _compare_once(r, o, col, flux_tol_scale=1.0, **kw):
    """`flux_tol_scale` loosens the flux tolerances for deliberately ill-conditioned settings
    (see test_gpu_fuzz.py); the opacity tolerances never move."""
    f = flux_tol_scale
    isr, olr = r.TOA_fluxes(*col.args(), **kw)
    isr_o, olr_o = o.TOA_fluxes(*col.args(), **kw)
    assert abs(olr - olr_o) <= f * RTOL_TOA * abs(olr_o)
    assert abs(isr - isr_o) <= f * RTOL_TOA * max(abs(isr_o), 1e-300)
    for wg, wo in ((r.wrk_ir, o.wrk_ir), (r.wrk_sol, o.wrk_sol)):
        # The up and down fluxes of a channel come out of one linear solve as sums of terms of the
        # size of the larger of the two (y1*e + y2*e + C, twostream.f90:143-148): where one of them
        # is orders of magnitude below the other (a single optically thin layer: fdn 7.5 against
        # fup 3e8 in fuzz case 51) an ulp of the large terms is 1e-8 of the small flux -- in the
        # reference's own arithmetic just as here.  So both are measured on their common scale.
        n_scale = max(float(np.max(np.abs(wo.fup_n))), float(np.max(np.abs(wo.fdn_n))), 1e-300)
        a_scale = max(float(np.max(np.abs(wo.fup_a))), float(np.max(np.abs(wo.fdn_a))), 1e-300)
        assert float(np.max(np.abs(np.asarray(wg.fup_n) - np.asarray(wo.fup_n)))) <= f * TOL_LEVEL * n_scale
        assert float(np.max(np.abs(np.asarray(wg.fdn_n) - np.asarray(wo.fdn_n)))) <= f * TOL_LEVEL * n_scale
        assert float(np.max(np.abs(np.asarray(wg.fup_a) - np.asarray(wo.fup_a)))) <= f * TOL_SPEC * a_scale
        assert float(np.max(np.abs(np.asarray(wg.fdn_a) - np.asarray(wo.fdn_a)))) <= f * TOL_SPEC * a_scale
        assert _scaled(wg.amean, wo.amean) <= f * TOL_SPEC
        assert _rel(wg.tau_band, wo.tau_band) <= RTOL_OPR
    assert _scaled(r.f_total, o.f_total) <= f * TOL_LEVEL
    for a, b in zip(r.opr(), o.opr()):
        assert _rel(a, b) <= RTOL_OPR
    return isr, olr


def test_hip_extension_is_loaded(hip_lib):
    """The tests below run the in-tree HIP library, not a fallback."""
    import os
    from clima_amd import lib
    maps = open("/proc/%d/maps" % os.getpid()).read()
    assert os.path.basename(lib.LIB_PATH) in maps


def test_device_exp(hip_lib):
    """The kernels' own exp (Cody-Waite + degree-13 polynomial) against libm: <= 2 ulp over the
    argument ranges of this path, exact limits at the ends."""
    import ctypes as C
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-760, 5, 200000), rng.uniform(-1e-3, 1e-3, 1000), -10 ** rng.uniform(-12, 7, 2000),
                        rng.uniform(600, 720, 1000), [0.0, -0.0, -745.2, -1e9, 709.7, 710.5, 709.78, 709.1, 1e6, -1e6, 5e9, -5e9, 1e14, -1e14]])
    y = np.empty_like(x)
    err = C.create_string_buffer(1025)
    dp = C.POINTER(C.c_double)
    hip_lib.clima_test_device_exp(C.byref(C.c_int(len(x))), x.ctypes.data_as(dp), y.ctypes.data_as(dp), err)
    assert err.value == b""
    with np.errstate(over="ignore", under="ignore"):
        ref = np.exp(x)
    normal = (ref > 1e-300) & np.isfinite(ref)
    ulp = np.abs(y[normal] - ref[normal]) / np.spacing(ref[normal])
    assert ulp.max() <= 2.0, ulp.max()
    assert np.all(y[ref == 0.0] == 0.0) and np.all(np.isinf(y[np.isinf(ref)]))
    assert np.all(np.abs(y[~normal & (ref > 0) & np.isfinite(ref)] - ref[~normal & (ref > 0) & np.isfinite(ref)]) <= 1e-300)


def test_device_exp_table(hip_lib):
    """The table exp of the solar zenith-angle loop (arguments <= 0): within 1.5 ulp + |x| * 2^-53 relative
    (the second term is the rounding of the scaled argument), 0 for anything below the underflow limit."""
    import ctypes as C
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-745, 0, 200000), -10 ** rng.uniform(-14, 3, 50000), rng.uniform(-1e-2, 0, 5000),
                        [0.0, -0.0, -745.2, -800.0, -1e6, -1e9, -2.3e7, -1e14, -1e300, -np.log(2) / 256, -np.log(2) / 512]])
    y = np.empty_like(x)
    err = C.create_string_buffer(1025)
    dp = C.POINTER(C.c_double)
    hip_lib.clima_test_device_exp_table(C.byref(C.c_int(len(x))), C.byref(C.c_int(0)), x.ctypes.data_as(dp), y.ctypes.data_as(dp), err)
    assert err.value == b""
    with np.errstate(under="ignore"):
        ref = np.exp(x)
    normal = ref > 1e-300
    rel = np.abs(y[normal] - ref[normal]) / ref[normal]
    assert np.all(rel <= (1.5 + np.abs(x[normal])) * 2.0 ** -52), (rel / ((1.5 + np.abs(x[normal])) * 2.0 ** -52)).max()
    assert np.all(y[ref == 0.0] == 0.0)
    assert np.all(np.abs(y[~normal] - ref[~normal]) <= 1e-300)
    assert y[-11] == 1.0 and y[-10] == 1.0        # exp(0), exp(-0)


def test_device_ten2power_table(hip_lib):
    """10^y of the opacity tile's table interpolations (log10 k, log10 sigma: -60 ... +5), table form:
    within 1.5 ulp + |y| ln(10) 2^-53 relative of the correctly rounded value."""
    import ctypes as C
    from decimal import Decimal, getcontext
    rng = np.random.default_rng(3)
    y = np.concatenate([rng.uniform(-60, 5, 200000), rng.uniform(-300, -60, 2000), rng.uniform(-1e-3, 1e-3, 1000), [0.0, -0.0, 1.0, -1.0, 2.0, -330.0, -400.0]])
    out = np.empty_like(y)
    err = C.create_string_buffer(1025)
    dp = C.POINTER(C.c_double)
    hip_lib.clima_test_device_exp_table(C.byref(C.c_int(len(y))), C.byref(C.c_int(1)), y.ctypes.data_as(dp), out.ctypes.data_as(dp), err)
    assert err.value == b""
    # reference: 10^y in extended precision for a sample, float64 power for the rest (itself good to < 1 ulp)
    with np.errstate(under="ignore"):
        ref = np.power(10.0, y)
    getcontext().prec = 40
    for i in rng.integers(0, 200000, 300):
        ref[i] = float(Decimal(10) ** Decimal(float(y[i])))
    normal = ref > 1e-300
    rel = np.abs(out[normal] - ref[normal]) / ref[normal]
    bound = (2.5 + np.abs(y[normal]) * np.log(10.0)) * 2.0 ** -52      # (+1 ulp for the float64 reference)
    assert np.all(rel <= bound), (rel / bound).max()
    assert np.all(out[ref == 0.0] == 0.0)
    assert out[-7] == 1.0 and out[-6] == 1.0 and abs(out[-5] - 10.0) <= 2e-15 * 10 and abs(out[-3] - 100.0) <= 1e-13


def test_device_rcp_and_sqrt(hip_lib):
    """The kernels' reciprocal (v_rcp_f64 + 2 Newton steps) and square root (v_rsq_f64 + Goldschmidt),
    used where the reference divides or calls sqrt on quantities of ordinary size: <= 1 ulp."""
    import ctypes as C
    rng = np.random.default_rng(1)
    x = np.concatenate([10.0 ** rng.uniform(-6, 6, 200000), rng.uniform(1.0, 4.0, 200000)])
    y = np.empty(4 * len(x))
    err = C.create_string_buffer(1025)
    dp = C.POINTER(C.c_double)
    hip_lib.clima_test_device_rcp(C.byref(C.c_int(len(x))), x.ctypes.data_as(dp), y.ctypes.data_as(dp), err)
    assert err.value == b""
    y = y.reshape(4, -1)
    for got, ref in ((y[2], 1.0 / x), (y[3], np.sqrt(x))):
        ulp = np.abs(got - ref) / np.spacing(ref)
        assert ulp.max() <= 1.0, ulp.max()


def test_dpp_wave_scans(hip_lib):
    # the DPP-based affine wave scan used by the batched IR kernel, against a serial recurrence
    import ctypes as C
    rng = np.random.default_rng(4)
    nw = 5
    a = rng.normal(size=nw * 64)
    b = rng.uniform(0.2, 1.1, size=nw * 64) * rng.choice([1.0, -1.0], size=nw * 64)
    out = np.empty(4 * nw * 64)
    err = C.create_string_buffer(1025)
    dp = C.POINTER(C.c_double)
    hip_lib.clima_test_wave_scan(C.byref(C.c_int(nw)), a.ctypes.data_as(dp), b.ctypes.data_as(dp), out.ctypes.data_as(dp), err)
    assert err.value == b""
    out = out.reshape(4, nw, 64)
    A, B = a.reshape(nw, 64), b.reshape(nw, 64)
    want = np.empty_like(A)
    for w in range(nw):
        x = 0.0
        for i in range(64):
            x = A[w, i] + B[w, i] * x
            want[w, i] = x
    np.testing.assert_allclose(out[0], want, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(out[1], want, rtol=1e-12, atol=1e-13)
    shifted = np.concatenate([np.zeros((nw, 1)), A[:, :-1]], axis=1)
    np.testing.assert_array_equal(out[2], shifted)
    np.testing.assert_array_equal(out[3], A[:, ::-1])


def test_config1_modern_earth_50_layers(O, small_tables):
    # BASELINE.json configs[0]: ModernEarth, 50 layers, 1 zenith angle, albedo 0.3
    from clima_amd import synthetic as S
    r, o = _pair(O, small_tables, 50, 1, 0.3)
    _compare(r, o, S.modern_earth_column(50))


@pytest.mark.parametrize("nz", [1, 2, 3, 5, 16, 63, 64, 65])
def test_layer_counts_and_chunk_edges(O, nz):
    # ragged sizes around the wave width and the chunked solve's boundaries
    from clima_amd import synthetic as S
    tb = S.modern_earth_tables(nw=10, seed=100 + nz)
    r, o = _pair(O, tb, nz, 2, 0.25)
    _compare(r, o, S.modern_earth_column(nz))


def test_multiple_zenith_angles_and_scalars(O, small_tables):
    from clima_amd import synthetic as S
    r, o = _pair(O, small_tables, 40, 8, 0.15, photon_scale_factor=0.4286, diurnal_fac=0.37)
    _compare(r, o, S.modern_earth_column(40))


@pytest.mark.parametrize("nz,nzen", [(40, 12), (100, 12), (100, 3), (200, 16), (150, 1), (300, 10), (450, 5)])
def test_zenith_counts_across_launch_forms(O, nz, nzen):
    # 1..16 zenith angles (the fused grid unrolls the first 8 and loops over the rest) on columns
    # that take the separate launches (40 layers), the fused grid with 2, 3 and 4 layer slots per
    # lane (100, 150, 200), and the 5- and 8-slot stand-alone kernels (300, 450)
    from clima_amd import synthetic as S
    tb = S.modern_earth_tables(nw=12, seed=500 + nz + nzen)
    r, o = _pair(O, tb, nz, nzen, 0.2, diurnal_fac=0.5)
    _compare(r, o, S.modern_earth_column(nz))


def test_no_hard_surface_and_emissivity(O, small_tables):
    from clima_amd import synthetic as S
    r, o = _pair(O, small_tables, 30, 2, 0.2, has_hard_surface=False, ir_tau_min=1e-3)
    _compare(r, o, S.modern_earth_column(30))
    r2, o2 = _pair(O, small_tables, 30, 2, 0.2)
    em = np.linspace(0.6, 1.0, len(r2.surface_emissivity))
    al = np.linspace(0.0, 0.9, len(r2.surface_albedo))
    r2.surface_emissivity = em
    r2.surface_albedo = al
    o2.set_surface_emissivity(em)
    o2.set_surface_albedo(al)
    _compare(r2, o2, S.modern_earth_column(30))


def test_pair_reuse_doubled_grid(O, small_tables):
    # AdiabatClimate's doubled radiative grid: identical layer pairs (types.f90:621-632)
    from clima_amd import synthetic as S
    col = S.doubled_column(S.modern_earth_column(32))
    r, o = _pair(O, small_tables, 64, 4, 0.3)
    _compare(r, o, col)
    # pairs equal only to 1e-13 still count as reusable; the second layer copies the first
    col2 = S.doubled_column(S.modern_earth_column(32))
    col2["T"][1::2] *= 1 + 1e-13
    _compare(r, o, col2)


def test_unsorted_k_coefficients_use_full_network(O):
    from clima_amd import synthetic as S
    tb = S.modern_earth_tables(nw=24, sorted_k=False, seed=11)
    r, o = _pair(O, tb, 31, 3, 0.1)
    _compare(r, o, S.modern_earth_column(31))


def test_uneven_g_weights_multi_edge_rebin(O):
    # split quadrature with tiny weights near g -> 1 (as real k-tables have): a sorted element can
    # then be wider than an output bin and cross several edges at once (max(wxy) > min(wbin))
    from clima_amd import synthetic as S
    w = np.array([0.30, 0.28, 0.20, 0.12, 0.06, 0.025, 0.011, 0.004])
    tb = S.modern_earth_tables(nw=16, seed=77, weights=w / w.sum())
    r, o = _pair(O, tb, 33, 2, 0.3)
    _compare(r, o, S.modern_earth_column(33))


@pytest.mark.parametrize("generic", [0, 1])
@pytest.mark.parametrize("ng,sorted_k", [(1, True), (4, True), (6, True), (12, False), (12, True), (16, True), (20, True), (24, False),
                                         (32, True)])
def test_other_g_point_counts(O, ng, sorted_k, generic, monkeypatch):
    # `new_num_k_bins` need not be 8.  generic = 0: the group-of-lanes kernel with the next power of two of lanes per
    # item, the lanes beyond ng padded (round 3: 12 g-points 4.2 ms -> 0.66 ms at config 2's size); generic = 1
    # (CLIMA_HIP_GENERIC): the wave-per-item resort-rebin kernel (bitonic sort on (value, index) in LDS, the
    # reference's arithmetic order) kept as a cross-check.  Both with the per-group two-stream launches.
    from clima_amd import synthetic as S
    tb = S.modern_earth_tables(nw=12, ng=ng, sorted_k=sorted_k, seed=5 + ng)
    monkeypatch.setenv("CLIMA_HIP_GENERIC", str(generic))
    r, o = _pair(O, tb, 22, 2, 0.25)
    monkeypatch.delenv("CLIMA_HIP_GENERIC")
    _compare(r, o, S.doubled_column(S.modern_earth_column(11)))


def _custom_props(nwv=7, nP=6, seed=3):
    rng = np.random.default_rng(seed)
    wv = np.geomspace(150.0, 4.0e5, nwv)             # nm; narrower than the grid -> constant extrapolation
    # dynes/cm^2, decreasing; the column (1e6 ... ~0.2) reaches a little beyond both ends, so the
    # end intervals extrapolate -- mildly, keeping w0 and g0 physical
    P = np.geomspace(0.6e6, 0.5, nP)
    dtau_dz = 10.0 ** (-8.0 + rng.uniform(-0.5, 0.5, (nP, nwv)))
    w0 = 0.5 + 0.2 * rng.uniform(-1.0, 1.0, (nP, nwv))
    g0 = 0.3 + 0.3 * rng.uniform(-1.0, 1.0, (nP, nwv))
    return wv, P, dtau_dz, w0, g0


@pytest.mark.parametrize("ng", [8, 4])
def test_custom_optical_properties(O, ng):
    # Radtran%set_custom_optical_properties (clima_radtran.f90:494-512): tuned and generic kernels
    from clima_amd import synthetic as S
    tb = S.modern_earth_tables(nw=20, ng=ng, seed=21)
    r, o = _pair(O, tb, 30, 2, 0.2)
    col = S.modern_earth_column(30)
    isr0, olr0 = _compare(r, o, col)
    args = _custom_props()
    r.set_custom_optical_properties(*args)
    o.set_custom_optical_properties(*args)
    isr1, olr1 = _compare(r, o, col)
    assert abs(olr1 - olr0) > 1e-3 * abs(olr0)      # the custom opacity matters in this case
    r.unset_custom_optical_properties()
    o.unset_custom_optical_properties()
    isr2, olr2 = _compare(r, o, col)
    assert (isr2, olr2) == (isr0, olr0)


@pytest.mark.parametrize("ng", [8, 16, 32])
def test_group_of_lanes_kernel_pairs_custom_and_single_species(O, ng):
    # k_opacity_coop<NG> on the paths its one lane per g-point changes: a doubled grid whose pairs are exact
    # copies (results stored twice), pairs equal only to 1e-13 (the second layer's own terms are evaluated:
    # types.f90:621-632 compares to 1e-12), custom optical properties, and a single k-species on a doubled
    # grid (no mixing step to copy: k of the source layer times the second layer's own column, :818)
    from clima_amd import synthetic as S
    tb = S.modern_earth_tables(nw=14, ng=ng, seed=23)
    r, o = _pair(O, tb, 40, 2, 0.25)
    r.coop_items = 1 << 30
    col = S.doubled_column(S.modern_earth_column(20))
    _compare_once(r, o, col)
    col2 = S.doubled_column(S.modern_earth_column(20))
    col2["T"][1::2] *= 1 + 1e-13
    col2["densities"] = np.asfortranarray(col2["densities"])
    col2["densities"][1::2, 1] *= 1 - 2e-13
    _compare_once(r, o, col2)
    args = _custom_props()
    r.set_custom_optical_properties(*args)
    o.set_custom_optical_properties(*args)
    _compare_once(r, o, col2)
    _compare_once(r, o, S.modern_earth_column(40))
    tb1 = S.make_tables(nw=10, ng=ng, k_species=("CO2",), seed=6)
    r1, o1 = _pair(O, tb1, 40, 2, 0.25)
    r1.coop_items = 1 << 30
    _compare_once(r1, o1, col2)


def test_custom_optical_properties_errors(small_tables):
    from clima_amd.radtran import Radtran, ClimaException
    r = Radtran(small_tables, 10, 1, 0.2)
    wv, P, t, w, g = _custom_props()
    cases = [((-wv, P, t, w, g), "All elements of `wv` must be larger than zero"),
             ((wv, -P, t, w, g), "All elements of `P` must be larger than zero"),
             ((wv, P[:-1], t, w, g), "`P` and `dtau_dz` have incompatible shapes"),
             ((wv[:-1], P, t, w, g), "`wv` and `dtau_dz` have incompatible shapes"),
             ((wv, P, t, w[:-1], g), "`P` and `w0` have incompatible shapes"),
             ((wv, P, t, w, g[:, :-1]), "`wv` and `g0` have incompatible shapes"),
             ((wv[::-1].copy(), P, t, w, g), "Interpolation error in `set_custom_optical_properties`"),
             ((wv, P[::-1].copy(), t, w, g), "Interpolation initialization error in `set_custom_optical_properties`")]
    for a, msg in cases:
        with pytest.raises(ClimaException) as e:
            r.set_custom_optical_properties(*a)
        assert str(e.value) == msg


def test_ties_and_zero_columns(O, small_tables):
    # a species with zero abundance gives 8-fold ties in every resort
    from clima_amd import synthetic as S
    col = S.modern_earth_column(20)
    col["densities"][:, 1] = 0.0      # CO2 absent
    col["densities"][:, 4] = 0.0      # O3 absent
    r, o = _pair(O, small_tables, 20, 2, 0.3)
    _compare(r, o, col)


def test_single_k_species_no_resort(O):
    # nk = 1: the RORR loop is never entered (types.f90:823)
    from clima_amd import synthetic as S
    tb = S.make_tables(nw=16, k_species=("H2O",), seed=5)
    r, o = _pair(O, tb, 25, 2, 0.3)
    _compare(r, o, S.modern_earth_column(25))


def test_early_mars_cia_heavy(O):
    # BASELINE.json configs[2] at reduced bin count
    from clima_amd import synthetic as S
    tb = S.early_mars_tables(nw=60)
    r, o = _pair(O, tb, 200, 4, 0.2, photon_scale_factor=0.4286)
    _compare(r, o, S.early_mars_column(200))


def test_config5_500_layers(O):
    # BASELINE.json configs[4] column shape (500 layers: 8 layers per lane in the two-stream kernel)
    from clima_amd import synthetic as S
    tb = S.modern_earth_tables(nw=20, seed=55)
    r, o = _pair(O, tb, 500, 4, 0.15)
    _compare(r, o, S.modern_earth_column(500))


def test_more_than_512_layers_falls_back_to_lds_kernel(O):
    # beyond 8 layers per lane the wave kernel declines and the workgroup-per-bin kernel runs;
    # beyond what one 160 KiB LDS image holds the library reports it (no silent truncation)
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran, ClimaException
    tb = S.modern_earth_tables(nw=8, seed=56)
    r, o = _pair(O, tb, 640, 2, 0.15)
    _compare(r, o, S.modern_earth_column(640))
    big = Radtran(tb, 5000, 1, 0.15)
    with pytest.raises(ClimaException, match="exceeds what the two-stream kernels can stage"):
        big.radiate(*S.modern_earth_column(5000).args())


def test_config4_perturbed_columns(O, small_tables):
    # BASELINE.json configs[3]: perturbed ModernEarth columns (T-P and mixing-ratio sweep),
    # here a handful of them through one handle, one after the other
    from clima_amd import synthetic as S
    r, o = _pair(O, small_tables, 50, 2, 0.15)
    for col in S.perturbed_columns(6, nz=50, seed=7):
        _compare(r, o, col)


def test_workgroup_per_bin_kernel_agrees(O, small_tables, monkeypatch):
    # the LDS-staged workgroup-per-bin two-stream kernel (fallback form) gives the same answers
    from clima_amd import synthetic as S
    monkeypatch.setenv("CLIMA_HIP_TS_MODE", "block")
    r, o = _pair(O, small_tables, 50, 4, 0.3)
    _compare(r, o, S.modern_earth_column(50))


@pytest.mark.parametrize("nz,nw", [(50, 40), (200, 60), (13, 7), (65, 20), (100, 30), (128, 12), (129, 12), (150, 20), (192, 9), (256, 10)])
def test_fused_and_separate_launch_forms(O, nz, nw, monkeypatch):
    # default: opacity + two-stream blocks in one grid (k_fused, block-to-block hand-off inside the
    # launch); radtran_fused_set(0) / CLIMA_HIP_FUSED=0: one launch per kernel.  Same opacities bit
    # for bit (same code), same fluxes to rounding, both within tolerance of the oracle.  The fused
    # grid is used for 65..256 layers, with 2, 3 or 4 layer slots per lane (each boundary is here);
    # shorter columns always take the separate launches.
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    tb = S.modern_earth_tables(nw=nw, seed=31)
    col = S.modern_earth_column(nz)
    r, o = _pair(O, tb, nz, 3, 0.2)
    r.coop_items = 0      # the lane-per-item opacity kernel, whatever the item count (else small calls take k_opacity_coop)
    assert r.fused
    _compare(r, o, col)
    opr_f = [a.copy() for a in r.opr()]
    flux_f = np.array(r.f_total)
    r.fused = False
    assert not r.fused
    _compare(r, o, col)
    for a, b in zip(opr_f, r.opr()):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_allclose(np.array(r.f_total), flux_f, rtol=1e-11, atol=1e-9 * np.max(np.abs(flux_f)))
    monkeypatch.setenv("CLIMA_HIP_FUSED", "0")
    assert not Radtran(tb, nz, 3, 0.2).fused


@pytest.mark.parametrize("nz,nw,doubled", [(50, 40, False), (200, 60, False), (13, 7, False), (64, 20, True), (102, 30, True),
                                            (300, 12, False), (402, 10, True)])
def test_group_of_lanes_opacity_kernel(O, nz, nw, doubled):
    # k_opacity_coop<8> (8 lanes per (bin, source layer): cross-lane bitonic sort over ds_swizzle, rebin on
    # the distributed sorted keys) -- what calls with few items use -- against the oracle and against the
    # lane-per-item kernel (same opacities to rounding: the running weights are summed in another order)
    from clima_amd import synthetic as S
    tb = S.modern_earth_tables(nw=nw, seed=17)
    col = S.modern_earth_column(nz)
    if doubled:
        col = S.doubled_column(S.modern_earth_column(nz // 2))
    r, o = _pair(O, tb, nz, 3, 0.2)
    r.coop_items = 1 << 30
    _compare(r, o, col)
    opr_c = [a.copy() for a in r.opr()]
    r.coop_items = 0
    _compare(r, o, col)
    for a, b in zip(opr_c, r.opr()):
        assert _rel(a, b) <= 1e-12


def test_fused_handoff_timeout_is_reissued_unfused(O, monkeypatch):
    """A two-stream block of the fused grid whose (bounded) wait for its opacity blocks expires must
    not fail the call with the opacity error text: the call is computed again through the separate
    launches.  CLIMA_HIP_FUSED_SPINS=0 makes every wait that is not satisfied at its first poll expire."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    tables = S.modern_earth_tables(nw=400)   # enough opacity blocks that some two-stream blocks do wait
    nz = 200
    col = S.modern_earth_column(nz)
    ref = Radtran(tables, nz, 4, 0.2)
    ref.fused = False
    want = ref.TOA_fluxes(*col.args())
    want_f, want_opr = np.array(ref.f_total), ref.opr()
    monkeypatch.setenv("CLIMA_HIP_FUSED_SPINS", "0")
    r = Radtran(tables, nz, 4, 0.2)
    monkeypatch.delenv("CLIMA_HIP_FUSED_SPINS")
    assert r.fused
    got = r.TOA_fluxes(*col.args())          # synchronous API: re-issued inside the call
    assert r.fused_fallbacks >= 1
    assert got == want
    np.testing.assert_array_equal(np.array(r.f_total), want_f)
    for a, b in zip(r.opr(), want_opr):
        np.testing.assert_array_equal(a, b)
    # resident form: detected at the synchronise
    n0 = r.fused_fallbacks
    r.upload_column(*col.args())
    r.radiate_resident()
    r.synchronize()
    assert r.fused_fallbacks > n0
    np.testing.assert_array_equal(np.array(r.f_total), want_f)
    # column batch
    n0 = r.fused_fallbacks
    isr, olr = r.TOA_fluxes_batch([col, col])
    assert r.fused_fallbacks > n0 and isr[0] == want[0] and olr[1] == want[1]


def test_state_carried_between_calls(O, small_tables):
    # compute_opacity=False reuses opr; compute_solar=False reuses wrk_sol (clima_radtran.f90:255-289)
    from clima_amd import synthetic as S
    col = S.modern_earth_column(50)
    r, o = _pair(O, small_tables, 50, 2, 0.3)
    _compare(r, o, col)
    warm = S.Column(col)
    warm["T"] = col["T"] + 3.0
    warm["T_surface"] = col["T_surface"] + 3.0
    _compare(r, o, warm, compute_solar=False, compute_opacity=False)   # the RCE-Jacobian call pattern
    _compare(r, o, warm, compute_solar=False)
    _compare(r, o, warm)


@pytest.mark.parametrize("nz,ncol", [(50, 7), (30, 70), (300, 5), (1, 3), (64, 9), (65, 9), (100, 5), (150, 9), (193, 4), (256, 3),
                                     (257, 4), (320, 9), (402, 11), (448, 3), (500, 6), (512, 3), (20, 600)])   # 600 columns: two launches of <= 512
def test_batched_shared_opacity_ir_calls(O, small_tables, nz, ncol, monkeypatch):
    # the RCE Jacobian's loop (clima_adiabat_solve.f90:798-812) in one call: every column equals
    # radiate(..., compute_solar=False, compute_opacity=False) on the same resident opacities
    from clima_amd import synthetic as S
    col = S.modern_earth_column(nz)
    r, o = _pair(O, small_tables, nz, 2, 0.3)
    _compare(r, o, col)
    rng = np.random.default_rng(9)
    T = np.repeat(np.asarray(col["T"])[:, None], ncol, axis=1)
    Ts = np.full(ncol, float(col["T_surface"]))
    for c in range(ncol):           # one perturbed level per column, as the Jacobian does
        k = c % (nz + 1)
        dT = 1.0e-2 * (1.0 + rng.random()) * (Ts[c] if k == 0 else T[k - 1, c])
        if k == 0:
            Ts[c] += dT
        else:
            T[k - 1, c] += dT
    base_up = np.array(r.wrk_ir.fup_n)
    fup, fdn, ftot = r.radiate_ir_batch(Ts, T)
    assert fup.shape == (nz + 1, ncol)
    np.testing.assert_array_equal(np.array(r.wrk_ir.fup_n), base_up)   # the handle's results stay
    for c in range(ncol):
        w = S.Column(col)
        w["T"] = T[:, c].copy()
        w["T_surface"] = Ts[c]
        o.radiate(*w.args(), compute_solar=False, compute_opacity=False)
        assert _scaled(fup[:, c], o.wrk_ir.fup_n) <= TOL_LEVEL
        assert _scaled(fdn[:, c], o.wrk_ir.fdn_n) <= TOL_LEVEL
        assert _scaled(ftot[:, c], o.f_total) <= TOL_LEVEL
    # against the one-at-a-time path of the library itself: the shared-matrix batch kernel
    # splits the arithmetic differently (rounding-level differences), the per-column form
    # (CLIMA_HIP_BATCH_SHARED=0) is the same arithmetic bit for bit
    k3 = min(3, ncol - 1)
    w = S.Column(col)
    w["T"] = T[:, k3].copy()
    w["T_surface"] = Ts[k3]
    r.radiate(*w.args(), compute_solar=False, compute_opacity=False)
    one_up, one_ft = np.array(r.wrk_ir.fup_n), np.array(r.f_total)
    np.testing.assert_allclose(fup[:, k3], one_up, rtol=1e-11)
    np.testing.assert_allclose(ftot[:, k3], one_ft, rtol=1e-11, atol=1e-11 * np.max(np.abs(one_ft)))
    monkeypatch.setenv("CLIMA_HIP_BATCH_SHARED", "0")
    from clima_amd.radtran import Radtran
    r2 = Radtran(small_tables, nz, 2, 0.3)
    r2.coop_items = 0      # the opacities resident in `r` are the lane-per-item kernel's (_compare's second pass)
    r2.radiate(*col.args())
    fup2, fdn2, ftot2 = r2.radiate_ir_batch(Ts, T)
    np.testing.assert_array_equal(fup2[:, k3], one_up)
    np.testing.assert_array_equal(ftot2[:, k3], one_ft)
    np.testing.assert_allclose(fup2, fup, rtol=1e-11)


def test_adiabat_style_doubled_radiative_grid(O, small_tables):
    # AdiabatClimate's RT grid (clima_adiabat.f90:728-771): nz_r = 2*nz + 2 with ghost layers;
    # every pair is a pair_reuse pair (clima_radtran_types.f90:621-632)
    from clima_amd import synthetic as S
    from clima_amd.atmosphere import copy_atm_to_radiative_grid
    col = S.Column(copy_atm_to_radiative_grid(S.modern_earth_column(24)))
    assert len(col["T"]) == 50
    r, o = _pair(O, small_tables, 50, 4, 0.2)
    _compare(r, o, col)


def test_constructed_from_a_data_directory(O, tmp_path):
    # the reference constructor's argument list: settings YAML + star file + data directory
    # (clima_radtran.f90:98-126), tables through clima_amd/data_loader.py
    import os
    from clima_amd import data_loader as D
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    from datadir_fixture import write_datadir
    root = str(tmp_path)
    write_datadir(root, S.modern_earth_tables(nw=24, seed=12))
    settings, star = os.path.join(root, "settings.yaml"), os.path.join(root, "star.txt")
    r = Radtran.from_files(settings, star, 3, 0.2, 40, root)
    o = O.OracleRadtran(D.load_tables(settings, star, root), 40, 3, 0.2)
    assert r.species_names == list(S.MODERN_EARTH_SPECIES) and r.particle_names == ["HCaer1"]
    _compare(r, o, S.modern_earth_column(40))


def test_constructed_from_the_c_written_data_directory(O):
    # tests/golden/datadir_c/: HDF5 files written by the HDF5 C library (chunked, deflate, float32 for
    # one k-table and the Mie tables; tests/golden/h5pack.c), NOT by clima_amd/h5lite.py.  The HIP path is
    # fed through the loader; the oracle gets tables built from the arrays that went into the files by
    # tests/expected_tables.py, without the loader.
    import os
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    from expected_tables import DATADIR_C, expected_tables
    r = Radtran.from_files(os.path.join(DATADIR_C, "settings.yaml"), os.path.join(DATADIR_C, "star.txt"), 3, 0.2, 40, DATADIR_C)
    o = O.OracleRadtran(expected_tables(), 40, 3, 0.2)
    assert r.species_names == list(S.MODERN_EARTH_SPECIES) and r.particle_names == ["HCaer1"]
    _compare(r, o, S.modern_earth_column(40))


def test_column_batch_equals_one_call_per_column(O, small_tables):
    # BASELINE config 4 mechanics: radtran_toa_fluxes_batch enqueues the same kernels per column
    # without host round trips, so every column equals its own TOA_fluxes call bit for bit
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    nz = 50
    cols = S.perturbed_columns(9, nz, seed=3)
    r = Radtran(small_tables, nz, 2, 0.3)
    isr, olr, fl = r.TOA_fluxes_batch(cols, return_fluxes=True)
    assert fl.shape == (nz + 1, 5, len(cols))
    last_ft = np.array(r.f_total)
    np.testing.assert_array_equal(last_ft, fl[:, 4, -1])        # the handle holds the last column
    for c, col in enumerate(cols):
        one = r.TOA_fluxes(*col.args())
        assert one == (isr[c], olr[c])
        np.testing.assert_array_equal(np.array(r.wrk_ir.fup_n), fl[:, 0, c])
        np.testing.assert_array_equal(np.array(r.wrk_sol.fdn_n), fl[:, 3, c])
        np.testing.assert_array_equal(np.array(r.f_total), fl[:, 4, c])
    o = O.OracleRadtran(small_tables, nz, 2, 0.3)
    for c in (0, 4, 8):
        isr_o, olr_o = o.TOA_fluxes(*cols[c].args())
        assert abs(olr[c] - olr_o) <= RTOL_TOA * abs(olr_o) and abs(isr[c] - isr_o) <= RTOL_TOA * abs(isr_o)


def test_radiation_enhancement_and_bolometric(O, small_tables):
    from clima_amd import synthetic as S
    col = S.modern_earth_column(50)
    r, o = _pair(O, small_tables, 50, 2, 0.3)
    r.radiate(*col.args())
    sol = r.wrk_sol
    fup_n, fdn_a = sol.fup_n, sol.fdn_a
    ir = r.wrk_ir
    r.apply_radiation_enhancement(1.7)
    assert np.allclose(r.wrk_sol.fup_n, fup_n * 1.7, rtol=1e-15)
    assert np.allclose(r.wrk_sol.fdn_a, fdn_a * 1.7, rtol=1e-15)
    s2 = r.wrk_sol
    assert np.allclose(r.f_total, (s2.fdn_n - s2.fup_n) + (ir.fdn_n - ir.fup_n), rtol=1e-15)
    flux = r.bolometric_flux()
    ps, fr = r.photons_sol, r.sol.freq
    assert abs(flux - np.sum(ps * (fr[:-1] - fr[1:])) / 1e3) <= 1e-12 * flux
    r.set_bolometric_flux(1000.0)
    assert abs(r.bolometric_flux() - 1000.0) < 1e-9
    assert abs(r.equilibrium_temperature(0.3) - (1000.0 * 0.7 / (4 * 5.670374419e-8)) ** 0.25) < 1e-9
    assert abs(r.skin_temperature(0.3) - r.equilibrium_temperature(0.3) * 0.5 ** 0.25) < 1e-9


def test_error_behaviour_matches_reference(small_tables):
    from clima_amd import synthetic as S
    from clima_amd.radtran import ClimaException, Radtran
    col = S.modern_earth_column(50)
    r = Radtran(small_tables, 50, 1, 0.3)
    with pytest.raises(ClimaException, match="The model contains particles"):
        r.radiate(col["T_surface"], col["T"], col["P"], col["densities"], col["dz"])
    with pytest.raises(ClimaException, match='"T" has the wrong input dimension.'):
        r.radiate(col["T_surface"], col["T"][:-1], col["P"], col["densities"], col["dz"], col["pdensities"],
                  col["radii"])
    with pytest.raises(ClimaException, match='"densities" has the wrong input dimension.'):
        r.radiate(col["T_surface"], col["T"], col["P"], col["densities"][:, :-1], col["dz"], col["pdensities"],
                  col["radii"])
    with pytest.raises(ClimaException, match="Both pdensities and radii must be arguments."):
        r.radiate(col["T_surface"], col["T"], col["P"], col["densities"], col["dz"], col["pdensities"], None)
    # check_dimensions_p reports the two arrays separately (clima_radtran.f90:446-463) ...
    with pytest.raises(ClimaException, match='"radii" has the wrong input dimension.'):
        r.radiate(col["T_surface"], col["T"], col["P"], col["densities"], col["dz"], col["pdensities"],
                  col["radii"][:-1])
    with pytest.raises(ClimaException, match='"pdensities" has the wrong input dimension.'):
        r.radiate(col["T_surface"], col["T"], col["P"], col["densities"], col["dz"], col["pdensities"][:-1],
                  col["radii"])
    # ... and after the gas arrays (:441-445)
    with pytest.raises(ClimaException, match='"dz" has the wrong input dimension.'):
        r.radiate(col["T_surface"], col["T"], col["P"], col["densities"], col["dz"][:-1], col["pdensities"],
                  col["radii"][:-1])
    r.has_hard_surface = False
    assert r.has_hard_surface is False
    r.has_hard_surface = True
    assert r.has_hard_surface is True
    # particle radius outside the Mie grid -> every bin flags ierr (types.f90:973-976, :773-776)
    with pytest.raises(ClimaException, match="Opacity computation failed in one or more wavelength bins."):
        r.radiate(col["T_surface"], col["T"], col["P"], col["densities"], col["dz"], col["pdensities"],
                  col["radii"] * 1e6)
    r.radiate(*col.args())   # and the handle recovers
    with pytest.raises(ClimaException, match="is the wrong size"):
        r.zenith_u = np.ones(3)


def test_opacities2yaml_names(small_tables):
    from clima_amd.radtran import Radtran
    y = Radtran(small_tables, 10, 1, 0.2).opacities2yaml()
    assert y.startswith("  k-method: RandomOverlapResortRebin\n  opacities:\n    k-distributions: [H2O, CO2, O2, O3, CH4]")
    assert "    CIA: [N2-N2, O2-O2, CO2-CO2, O2-N2, CH4-CH4, CO2-CH4]" in y
    assert "    water-continuum: MT_CKD" in y and "particle-xs: [{name: HCaer1, data: " in y


def test_accessor_shapes_and_channels(small_tables):
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    r = Radtran(small_tables, 50, 4, 0.3)
    r.radiate(*S.modern_earth_column(50).args())
    nw_ir, nw_sol = len(small_tables.ir_wavl) - 1, len(small_tables.sol_wavl) - 1
    assert r.wrk_ir.fup_a.shape == (51, nw_ir) and r.wrk_ir.fup_a.flags.f_contiguous
    assert r.wrk_sol.amean.shape == (51, nw_sol) and r.wrk_sol.tau_band.shape == (50, nw_sol)
    assert np.all(r.wrk_ir.amean == 0.0)                                   # clima_radtran.f90:205
    assert np.array_equal(r.ir.wavl, small_tables.ir_wavl) and len(r.sol.freq) == nw_sol + 1
    assert np.allclose(r.ir.freq, 299792458.0 / (r.ir.wavl * 1e-9), rtol=1e-15)
    u = r.zenith_u
    assert len(u) == 4 and np.all((u > 0) & (u < 1)) and abs(np.sum(r.zenith_weights) - 1) < 1e-14
    assert r.f_total.shape == (51,)


def test_results_read_through_getters_after_an_unsynchronised_timed_out_call(monkeypatch):
    """radiate_resident + a spectra getter, with no radtran_synchronize in between: the getter itself must
    notice the expired hand-off and hand out the repaired results (round 2 copied out stale spectra)."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    tables = S.modern_earth_tables(nw=400)
    nz = 200
    col = S.modern_earth_column(nz)
    ref = Radtran(tables, nz, 4, 0.2)
    ref.fused = False
    ref.radiate(*col.args())
    want_up, want_tb = np.array(ref.wrk_sol.fup_a), np.array(ref.wrk_ir.tau_band)
    want_opr = ref.opr()
    monkeypatch.setenv("CLIMA_HIP_FUSED_SPINS", "0")
    r = Radtran(tables, nz, 4, 0.2)
    monkeypatch.delenv("CLIMA_HIP_FUSED_SPINS")
    other = S.modern_earth_column(nz)
    other["T"] = np.asarray(other["T"]) + 7.0
    r.radiate(*other.args())                      # something else in the buffers first
    n0 = r.fused_fallbacks
    r.upload_column(*col.args())
    r.radiate_resident()
    got_up = np.array(r.wrk_sol.fup_a)            # first touch after the call: a 2-D getter
    assert r.fused_fallbacks > n0
    np.testing.assert_array_equal(got_up, want_up)
    np.testing.assert_array_equal(np.array(r.wrk_ir.tau_band), want_tb)
    r.radiate_resident()
    for a, b in zip(r.opr(), want_opr):           # and radtran_opr_get
        np.testing.assert_array_equal(a, b)


def test_timed_out_opacity_pass_is_not_repaired_from_a_replaced_column(monkeypatch):
    """upload A, opacity pass (times out), upload B, compute_opacity=False pass, synchronize: recomputing the
    opacities from column B is not what was asked for -- the library says so instead."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran, ClimaException
    tables = S.modern_earth_tables(nw=400)
    nz = 200
    col = S.modern_earth_column(nz)
    monkeypatch.setenv("CLIMA_HIP_FUSED_SPINS", "0")
    r = Radtran(tables, nz, 4, 0.2)
    monkeypatch.delenv("CLIMA_HIP_FUSED_SPINS")
    r.upload_column(*col.args())
    r.radiate_resident()
    warm = S.Column(col)
    warm["T"] = np.asarray(col["T"]) + 3.0
    r.upload_column(*warm.args())
    r.radiate_resident(False, False)
    with pytest.raises(ClimaException, match="column has been replaced"):
        r.synchronize()
    # the caller repeats the steps: a fresh opacity pass clears the condition
    r.upload_column(*col.args())
    r.radiate_resident()
    r.synchronize()


def test_state_after_a_column_batch_is_the_last_columns(O):
    """After radtran_toa_fluxes_batch the handle's level rows AND its per-bin spectra are the last column's
    (round 2 left the spectra of an earlier call beside the new level rows)."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    tables = S.modern_earth_tables(nw=60)
    nz = 100
    cols = S.perturbed_columns(5, nz=nz, seed=3)
    r = Radtran(tables, nz, 2, 0.2)
    r.radiate(*cols[0].args())
    isr, olr = r.TOA_fluxes_batch(cols)
    up_n, up_a, tb = np.array(r.wrk_ir.fup_n), np.array(r.wrk_ir.fup_a), np.array(r.wrk_sol.tau_band)
    am = np.array(r.wrk_sol.amean)
    one = Radtran(tables, nz, 2, 0.2)
    one.coop_items = 0     # the batch runs the lane-per-item opacity kernel inside the fused grid: so does this call
    assert one.TOA_fluxes(*cols[-1].args()) == (isr[-1], olr[-1])
    np.testing.assert_array_equal(up_n, np.array(one.wrk_ir.fup_n))
    np.testing.assert_array_equal(up_a, np.array(one.wrk_ir.fup_a))
    np.testing.assert_array_equal(tb, np.array(one.wrk_sol.tau_band))
    np.testing.assert_array_equal(am, np.array(one.wrk_sol.amean))


@pytest.mark.parametrize("ng,nz", [(4, 50), (12, 70), (16, 130), (32, 40)])
def test_batched_shared_opacity_ir_calls_at_other_g_point_counts(O, ng, nz):
    """radtran_radiate_ir_batch with a k-distribution setting other than 8 g-points: round 2 fell back to one full
    solve per column; the shared-matrix kernel now takes the g-points in groups of four (k_twostream_ir_batch<L, 4>)."""
    from clima_amd import synthetic as S
    tb = S.modern_earth_tables(nw=12, ng=ng, seed=40 + ng)
    col = S.modern_earth_column(nz)
    r, o = _pair(O, tb, nz, 2, 0.25)
    r.radiate(*col.args())
    o.radiate(*col.args())
    ncol = 5
    T = np.repeat(np.asarray(col["T"])[:, None], ncol, axis=1)
    Ts = np.full(ncol, float(col["T_surface"]))
    Ts[1] += 2.0
    for c in range(2, ncol):
        T[(7 * c) % nz, c] += 3.0
    fup, fdn, ftot = r.radiate_ir_batch(Ts, T)
    for c in range(ncol):
        w = S.Column(col)
        w["T"] = T[:, c].copy()
        w["T_surface"] = Ts[c]
        o.radiate(*w.args(), compute_solar=False, compute_opacity=False)
        assert _scaled(fup[:, c], o.wrk_ir.fup_n) <= TOL_LEVEL
        assert _scaled(fdn[:, c], o.wrk_ir.fdn_n) <= TOL_LEVEL
        assert _scaled(ftot[:, c], o.f_total) <= TOL_LEVEL


@pytest.mark.parametrize("ng,nz", [(16, 100), (16, 200), (24, 70), (32, 130), (32, 224)])
def test_half_wave_two_stream_launch_at_16_24_32_g_points(O, ng, nz):
    """k_twostream_h<L>: the half-wave two-stream form as a launch of its own (8 g-point columns per block, one or two
    groups of 8 per launch), what calls with 16 / 24 / 32 g-points at 65-224 layers run (CLIMA_HIP_NO_HALF=1 selects
    the whole-wave kernel it replaces): against the oracle, and repeatable bit for bit."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    tb = S.modern_earth_tables(nw=10, ng=ng, seed=60 + ng)
    col = S.modern_earth_column(nz)
    r, o = _pair(O, tb, nz, 3, 0.2)
    _compare_once(r, o, col)
    half_up, half_ft = np.array(r.wrk_sol.fup_n), np.array(r.f_total)
    r2 = Radtran(tb, nz, 3, 0.2)
    r2.radiate(*col.args())
    np.testing.assert_array_equal(np.array(r2.f_total), half_ft)            # repeatable bit for bit (atomics of two addends)
    np.testing.assert_array_equal(np.array(r2.wrk_sol.fup_n), half_up)


def test_mixing_step_with_every_kind_of_wave(O):
    """The assembly form of the mixing step (clima_amd/csrc/rorr_xys_asm.inc; reference k_rorr,
    /root/reference/src/radtran/clima_radtran_types.f90:823-852) decides per 64-lane wave what it may leave out.  The
    k-tables here are built so that one call meets every case: a species whose whole range lies below the smallest gap of
    the mixture (row view, every merge left out, every row rebinned in closed form: the sums are already in the
    reference's order), a dominant species with wide gaps over a narrow mixture (column view), species of comparable
    size (nothing left out), bins where only the top rows stand alone (some merges, some rows), and -- second handle --
    g-points in scrambled order (the general network).  All against the oracle at the usual 1e-11 on tau.
    The closed-form rows differ from the sorted path at the 1e-15 level: which is why this is a
    tolerance test and not a bitwise one."""
    from clima_amd import synthetic as S
    nz = 70
    for sorted_k in (True, False):
        tb = S.make_tables(nw=24, seed=77, sorted_k=sorted_k)
        ng = tb.ng
        g = np.arange(ng, dtype=float)
        for bi in range(tb.nw):
            kind = bi % 4
            for si, k in enumerate(tb.ktables):
                a = k["log10k"]                      # [bin][T][P][g]
                smooth = 0.002 * (k["temp"][:, None] - 300.0) + 0.1 * (k["log10P"][None, :] + 2.0)
                col_scale = -np.log10({"H2O": 5e22, "CO2": 8e21, "O2": 4.5e24, "O3": 1e19, "CH4": 4e19}[tb.species_names[k["sp_ind"]]])
                if kind == 0:      # species 0 wide gaps; the others tiny and nearly flat: rows never interleave
                    base, ramp = (-1.0, 1.0) if si == 0 else (-9.0 - si, 0.004)
                elif kind == 1:    # species 1 dominant with wide gaps over a narrow mixture: columns never interleave
                    base, ramp = (-6.0, 0.01) if si == 0 else ((0.0, 1.1) if si == 1 else (-9.0 - si, 0.003))
                elif kind == 2:    # comparable sizes: everything interleaves
                    base, ramp = -2.0 + 0.1 * si, 0.35
                else:              # steep tails: the top rows peel off, the bottom ones interleave
                    base, ramp = -3.0 + 0.05 * si, 0.0
                vals = base + ramp * g + (0.0 if kind != 3 else 0.02 * g + 0.9 * np.maximum(g - 4.0, 0.0) ** 1.5)
                a[bi] = col_scale - 2.0 + smooth[:, :, None] + vals[None, None, :]
            if not sorted_k:
                for k in tb.ktables:
                    k["log10k"][bi] = k["log10k"][bi][..., ::-1] if bi % 2 else np.roll(k["log10k"][bi], 3, axis=-1)
        r, o = _pair(O, tb, nz, 2, 0.2)
        r.coop_items = 0          # the lane-per-item tile (the assembly block) whatever the item count
        _compare_once(r, o, S.modern_earth_column(nz))
