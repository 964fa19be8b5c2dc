"""BASELINE.json's full-size configuration (200 layers, 1000 bins, 8 g-points, 8 zenith
angles) -- checked against the oracle once, and through size-independent properties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nominal():
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    tb = S.modern_earth_tables()
    return tb, Radtran(tb, 200, 8, 0.15), S.modern_earth_column(200)


def test_config2_olr_against_oracle(O, nominal):
    tb, r, col = nominal
    o = O.OracleRadtran(tb, 200, 8, 0.15)
    isr_o, olr_o = o.TOA_fluxes(*col.args())
    isr, olr = r.TOA_fluxes(*col.args())
    assert abs(olr - olr_o) <= 1e-9 * abs(olr_o)          # north_star: 1e-4
    assert abs(isr - isr_o) <= 1e-9 * abs(isr_o)
    albedo = r.wrk_sol.fup_n[-1] / r.wrk_sol.fdn_n[-1]
    albedo_o = o.wrk_sol.fup_n[-1] / o.wrk_sol.fdn_n[-1]
    assert abs(albedo - albedo_o) <= 1e-9 * albedo_o
    for wg, wo in ((r.wrk_ir, o.wrk_ir), (r.wrk_sol, o.wrk_sol)):
        for a, b in ((wg.fup_n, wo.fup_n), (wg.fdn_n, wo.fdn_n)):
            assert np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b))
        for a, b in ((wg.fup_a, wo.fup_a), (wg.fdn_a, wo.fdn_a), (wg.amean, wo.amean)):
            assert np.max(np.abs(a - b)) <= 1e-8 * max(np.max(np.abs(b)), 1e-300)


def test_solar_is_linear_in_the_stellar_flux(nominal):
    tb, r, col = nominal
    r.photon_scale_factor = 1.0
    r.radiate(*col.args())
    base_up, base_dn, ir_up = r.wrk_sol.fup_n, r.wrk_sol.fdn_n, r.wrk_ir.fup_n
    r.photon_scale_factor = 0.25
    r.radiate(*col.args(), compute_opacity=False)
    assert np.allclose(r.wrk_sol.fup_n, 0.25 * base_up, rtol=1e-13)
    assert np.allclose(r.wrk_sol.fdn_n, 0.25 * base_dn, rtol=1e-13)
    # the IR does not see the star (to rounding: the call without the opacity step takes the stand-alone
    # two-stream kernel, whose lanes cut the column into other chunks than the fused grid's)
    assert np.allclose(r.wrk_ir.fup_n, ir_up, rtol=1e-12, atol=0.0)
    r.photon_scale_factor = 1.0


def test_integrals_are_sums_of_the_spectra(nominal):
    tb, r, col = nominal
    r.radiate(*col.args())
    for w, ch in ((r.wrk_ir, r.ir), (r.wrk_sol, r.sol)):
        dfreq = ch.freq[:-1] - ch.freq[1:]
        assert np.allclose(w.fup_n, w.fup_a @ dfreq, rtol=1e-12)
        assert np.allclose(w.fdn_n, w.fdn_a @ dfreq, rtol=1e-12)
    f = r.f_total
    assert np.allclose(f, (r.wrk_sol.fdn_n - r.wrk_sol.fup_n) + (r.wrk_ir.fdn_n - r.wrk_ir.fup_n), rtol=1e-13,
                       atol=1e-9)


def test_energy_conservation_of_conservative_limits(nominal):
    tb, r, col = nominal
    r.radiate(*col.args())
    sol = r.wrk_sol
    # net solar flux decreases monotonically downward (absorption only removes energy)
    net = sol.fdn_n - sol.fup_n
    assert np.all(np.diff(net) >= -1e-9 * net[-1])
    # nothing comes down in the thermal at the top; up-flux at the ground is the Planck surface term
    assert r.wrk_ir.fdn_n[-1] == 0.0
    assert 0 < sol.fup_n[-1] < sol.fdn_n[-1]


def test_repeatability_and_resident_path(nominal):
    tb, r, col = nominal
    a = r.TOA_fluxes(*col.args())
    b = r.TOA_fluxes(*col.args())
    assert a == b                                           # bitwise repeatable
    r.upload_column(*col.args())
    r.radiate_resident()
    r.synchronize()
    assert -(r.wrk_ir.fdn_n[-1] - r.wrk_ir.fup_n[-1]) == a[1]


def test_bin_sharded_partials_add_up(nominal):
    """config 5 mechanics on one GPU: every shard's partial level fluxes sum to the whole."""
    from clima_amd.sharding import bin_shard
    tb, r, col = nominal
    r.radiate(*col.args())
    full = np.stack([r.wrk_ir.fup_n, r.wrk_ir.fdn_n, r.wrk_sol.fup_n, r.wrk_sol.fdn_n])
    world = 4
    acc = np.zeros_like(full)
    ir0 = int(np.argmin(np.abs(tb.wavl - tb.ir_wavl[0])))
    nw_ir, nw_sol = len(tb.ir_wavl) - 1, len(tb.sol_wavl) - 1
    covered = 0
    for rank in range(world):
        r.set_bin_shard(rank, world)
        assert r.bin_shard() == bin_shard(tb.nw, (ir0, ir0 + nw_ir - 1), (0, nw_sol - 1), 8, rank, world)
        covered += r.bin_shard()[1]
        r.upload_column(*col.args())
        r.radiate_resident()
        r.synchronize()
        acc += np.stack([r.wrk_ir.fup_n, r.wrk_ir.fdn_n, r.wrk_sol.fup_n, r.wrk_sol.fdn_n])
    r.set_bin_shard(0, 1)
    assert covered == tb.nw
    # a quarter of the bins is few enough items for the group-of-lanes opacity kernel (k_opacity_coop<8>),
    # the whole grid runs the lane-per-item kernel: the two agree to rounding (6e-14 in tau), not bit for bit
    assert np.allclose(acc, full, rtol=1e-10, atol=1e-9)


def test_bin_sharded_ir_only_step_does_not_recount_solar(small_tables):
    """The RCE-Jacobian pattern on a bin-sharded handle: a full step, then `compute_solar=False`
    steps (clima_radtran.f90:286-289 keeps the last solar results).  The level-flux buffer is
    all-reduced IN PLACE, so after a step it holds reduced rows; an IR-only step must put this rank's
    PARTIAL solar rows back before the next reduce, or they are counted `world` times.  World 2 is
    emulated on one GPU: two sharded handles, the 'all-reduce' is their sum written into both."""
    import torch
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    tb = small_tables
    nz = 60
    col = S.modern_earth_column(nz)
    col2 = S.perturbed_columns(1, nz, seed=3)[0]
    ranks = [Radtran(tb, nz, 3, 0.2) for _ in range(2)]
    whole = Radtran(tb, nz, 3, 0.2)
    for k, r in enumerate(ranks):
        r.set_bin_shard(k, 2)
    flux = [r.flux_tensor() for r in ranks]

    def step(c, **kw):
        for r in ranks:
            r.upload_column(*c.args())
            r.radiate_resident(**kw)
            r.synchronize()
        total = flux[0] + flux[1]
        torch.cuda.synchronize()
        for f, r in zip(flux, ranks):
            f.copy_(total)
            torch.cuda.synchronize()
            r.finish_reduced()
        whole.radiate(*c.args(), **kw)
        for r in ranks:
            np.testing.assert_allclose(np.array(r.f_total), np.array(whole.f_total), rtol=1e-12, atol=1e-9)
            np.testing.assert_allclose(r.wrk_sol.fdn_n, whole.wrk_sol.fdn_n, rtol=1e-12, atol=1e-9)

    step(col)
    step(col2, compute_solar=False)
    step(col, compute_solar=False, compute_opacity=False)
    step(col2)


def test_single_scattering_albedo_left_to_the_fused_grid_is_materialised_on_demand(nominal):
    """The fused grid's opacity tiles do not write w0 (its two-stream part forms it from the layers' scattering
    optical depth); whoever asks for it afterwards -- the optical-property accessor, an IR-only call on the
    stored opacities -- gets the array formed by the expression of the tile's store: bitwise what the
    separate launches (which do write it) leave.
    NOTE what this does and does not say: the materialised array is `min(0.99999, scat / tau)` by true division
    (the reference's expression, types.f90:869-875), whereas the fused solve that ran BEFORE it was asked for
    used `scat * rcp(tau)` with the correctly rounded reciprocal -- up to 1 ulp from that quotient
    (kernels.hip twostream_p_body, `w0_from_scat`).  The exposed w0 is therefore the reference's, not bit for
    bit the one the fused two-stream solve consumed; the fluxes of the two launch forms agree to 1e-11
    (tests/test_gpu_parity.py::_compare runs both)."""
    tb, r, col = nominal
    assert r.fused
    r.radiate(*col.args())
    tau_f, w0_f, g_f, tb_f = r.opr()
    r.radiate(*col.args(), compute_solar=False, compute_opacity=False)     # two-stream kernels on the stored arrays
    ir_stored = np.array(r.wrk_ir.fup_n)
    r.fused = False
    try:
        r.radiate(*col.args())
        tau_u, w0_u, g_u, tb_u = r.opr()
        r.radiate(*col.args(), compute_solar=False, compute_opacity=False)
        ir_unfused = np.array(r.wrk_ir.fup_n)
    finally:
        r.fused = True
    np.testing.assert_array_equal(tau_f, tau_u)
    np.testing.assert_array_equal(w0_f, w0_u)
    np.testing.assert_array_equal(g_f, g_u)
    np.testing.assert_array_equal(ir_stored, ir_unfused)
    assert 0.0 < w0_f.max() <= 0.99999


def test_separate_launch_form_at_full_size(nominal):
    """config 2 with one launch per kernel (the form the fused grid replaces) gives the same TOA
    fluxes to rounding."""
    tb, r, col = nominal
    r.fused = True
    a = r.TOA_fluxes(*col.args())
    r.fused = False
    b = r.TOA_fluxes(*col.args())
    r.fused = True
    assert abs(a[0] - b[0]) <= 1e-12 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-12 * abs(a[1])


@pytest.mark.parametrize("nz", [200, 100, 150])
def test_fused_handoff_is_fresh_under_alternating_columns(nominal, nz):
    """The fused grid hands the opacities from producer to consumer blocks inside one launch
    (device-scope stores / loads, no cache-wide fences).  A stale read would return the PREVIOUS
    call's opacities, so alternate between different columns: every call must reproduce the
    separate-launch results of its own column (opacities bit for bit, fluxes to rounding).
    nz = 200, 100, 150: the fused grid's two-stream part with 4, 2 and 3 layer slots per lane."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    tb, r, col = nominal
    if nz != 200:
        r, col = Radtran(tb, nz, 8, 0.15), S.modern_earth_column(nz)
    cols = [col] + S.perturbed_columns(2, nz, seed=11)

    def run(c):
        r.upload_column(*c.args())
        r.radiate_resident()
        r.synchronize()
        return np.array(r.f_total), np.array(r.wrk_sol.fup_n), np.array(r.wrk_ir.fup_n)

    r.fused = False
    ref = [run(c) + tuple(a.copy() for a in r.opr()) for c in cols]
    r.fused = True
    for i in range(240):
        k = i % len(cols)
        got = run(cols[k])
        for a, b in zip(got, ref[k][:3]):
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-10 * np.max(np.abs(b)))
        if i % 40 < len(cols):
            for a, b in zip(r.opr(), ref[k][3:]):
                np.testing.assert_array_equal(a, b)


def _full_size_against_oracle(O, tb, nz, nzen, albedo, col, **scalars):
    from clima_amd.radtran import Radtran
    r = Radtran(tb, nz, nzen, albedo)
    o = O.OracleRadtran(tb, nz, nzen, albedo)
    for k, v in scalars.items():
        setattr(r, k, v)
    if scalars:
        o.set_scalars(**scalars)
    isr, olr = r.TOA_fluxes(*col.args())
    isr_o, olr_o = o.TOA_fluxes(*col.args())
    assert abs(olr - olr_o) <= 1e-9 * abs(olr_o) and abs(isr - isr_o) <= 1e-9 * abs(isr_o)
    for wg, wo in ((r.wrk_ir, o.wrk_ir), (r.wrk_sol, o.wrk_sol)):
        for a, b in ((wg.fup_n, wo.fup_n), (wg.fdn_n, wo.fdn_n)):
            assert np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b))
    assert np.max(np.abs(np.array(r.f_total) - np.array(o.f_total))) <= 1e-9 * np.max(np.abs(o.f_total))
    for a, b in zip(r.opr(), o.opr()):
        assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)) <= 1e-11


def test_config3_early_mars_full_size(O):
    """BASELINE.json configs[2]: CO2-dominated, CIA-heavy, 200 layers, 1000 bins, 4 zenith angles,
    photon scale factor of Mars (templates/AdiabatClimate/Mars/settings.yaml)."""
    from clima_amd import synthetic as S
    _full_size_against_oracle(O, S.early_mars_tables(), 200, 4, 0.2, S.early_mars_column(200),
                              photon_scale_factor=0.4286)


def test_config5_500_layers_full_size(O):
    """BASELINE.json configs[4] column shape at the full bin count (the bins are what an 8-GPU run
    shards; test_bin_sharded_partials_add_up covers the sharding itself)."""
    from clima_amd import synthetic as S
    _full_size_against_oracle(O, S.modern_earth_tables(), 500, 8, 0.15, S.modern_earth_column(500))


def test_config4_1024_perturbed_columns_at_full_size(O, nominal):
    """BASELINE.json configs[3] at its stated size: 1024 perturbed ModernEarth columns (SURVEY 8(d):
    whole-column dT ~ U(-20,20) + per-layer N(0,2 K), P x U(0.5,2), H2O and CO2 x 10^U(-1,1), seed 7),
    200 layers, the full 1000-bin grid, 8 zenith angles, through radtran_toa_fluxes_batch.
    A seeded subset of 16 columns is held to the oracle at the usual tolerances (level fluxes of both
    channels, ISR, OLR); every column is held to the one-call-per-column path bit for bit."""
    from clima_amd import synthetic as S
    tb, r, _ = nominal
    cols = S.perturbed_columns(1024, nz=200, seed=7)
    isr, olr, fl = r.TOA_fluxes_batch(cols, return_fluxes=True)        # fl (nz+1, 5, ncol)
    assert isr.shape == (1024,) and fl.shape == (201, 5, 1024)
    assert np.all(np.isfinite(fl)) and np.all(olr > 0) and np.all(isr > 0)
    assert np.ptp(olr) > 0.05 * np.mean(olr)                             # the sweep does move the answer
    # ---- bitwise: batch == single calls, all 1024
    for c, col in enumerate(cols):
        a = r.TOA_fluxes(*col.args())
        assert a == (isr[c], olr[c]), c
        if c % 64 == 0:
            np.testing.assert_array_equal(fl[:, 4, c], np.array(r.f_total))
            np.testing.assert_array_equal(fl[:, 0, c], r.wrk_ir.fup_n)
            np.testing.assert_array_equal(fl[:, 3, c], r.wrk_sol.fdn_n)
    # ---- oracle: 16 seeded columns
    o = O.OracleRadtran(tb, 200, 8, 0.15)
    pick = np.random.default_rng(7).choice(1024, size=16, replace=False)
    for c in pick:
        isr_o, olr_o = o.TOA_fluxes(*cols[c].args())
        assert abs(olr[c] - olr_o) <= 1e-9 * abs(olr_o), c               # north_star: 1e-4
        assert abs(isr[c] - isr_o) <= 1e-9 * abs(isr_o), c
        rows = (o.wrk_ir.fup_n, o.wrk_ir.fdn_n, o.wrk_sol.fup_n, o.wrk_sol.fdn_n, o.f_total)
        for pair in ((0, 1), (2, 3)):
            scale = max(np.max(np.abs(rows[pair[0]])), np.max(np.abs(rows[pair[1]])))
            for a in pair:
                assert np.max(np.abs(fl[:, a, c] - rows[a])) <= 1e-9 * scale, (c, a)
        assert np.max(np.abs(fl[:, 4, c] - rows[4])) <= 1e-9 * np.max(np.abs(rows[4])), c


@pytest.mark.parametrize("nz_adiabat,nw", [(50, 700), (100, 400), (150, 260), (200, 200)])
def test_doubled_radiative_grid_takes_the_paired_two_stream_form(O, nz_adiabat, nw):
    """AdiabatClimate's radiative grid (copy_atm_to_radiative_grid, src/adiabat/clima_adiabat.f90:729-773:
    nz_r = 2 nz + 2, every pair of layers identical) with enough bins for the fused grid: the opacity lanes
    exist per source layer only, and the two-stream part runs in its half-wave form up to 224 layers (102, 202: faster
    there than the paired form, round 3) and in its paired form beyond (302, 402 layers: 6, 8 slots, coefficients
    computed once per pair).  Against the oracle at the usual tolerances."""
    from clima_amd import synthetic as S
    from clima_amd.atmosphere import copy_atm_to_radiative_grid
    from clima_amd.radtran import Radtran
    from test_gpu_parity import _compare_once
    tb = S.modern_earth_tables(nw=nw, seed=5)
    col = S.Column(copy_atm_to_radiative_grid(S.modern_earth_column(nz_adiabat)))
    nzr = len(col["T"])
    assert nzr == 2 * nz_adiabat + 2
    r = Radtran(tb, nzr, 4, 0.2)
    assert r.nw * (nzr // 2) > r.coop_items          # the fused grid, not the small-call kernels
    o = O.OracleRadtran(tb, nzr, 4, 0.2)
    _compare_once(r, o, col)
    # the same column with one layer nudged (no longer all pairs): the unpaired form; both agree with the
    # oracle, and on the unchanged layers with each other to rounding
    f_pair = np.array(r.f_total)
    col2 = S.Column(col)
    col2["T"] = np.array(col["T"], copy=True)
    col2["T"][nzr // 2] *= 1.0 + 1.0e-9
    _compare_once(r, o, col2)
    np.testing.assert_allclose(np.array(r.f_total), f_pair, rtol=1e-6, atol=1e-9 * np.max(np.abs(f_pair)))


def test_rce_jacobian_batch_on_the_402_layer_doubled_grid_at_full_size(O):
    """The RCE Jacobian's radiative work at AdiabatClimate's default resolution (nz = 200 -> 402-layer doubled
    radiative grid, src/adiabat/clima_adiabat.f90:729-773; nz_r + 1 IR-only calls on unchanged opacities,
    clima_adiabat_solve.f90:768-822) through radtran_radiate_ir_batch at config 2's spectral size: every one of the
    403 columns against the library's own one-at-a-time IR-only call, three of them against the oracle."""
    from clima_amd import synthetic as S
    from clima_amd.atmosphere import copy_atm_to_radiative_grid
    from clima_amd.radtran import Radtran
    tb = S.modern_earth_tables()
    col = S.Column(copy_atm_to_radiative_grid(S.modern_earth_column(200)))
    nz = len(col["T"])
    assert nz == 402
    r = Radtran(tb, nz, 4, 0.15)
    r.radiate(*col.args())
    ncol = nz + 1
    T = np.repeat(np.asarray(col["T"])[:, None], ncol, axis=1)
    Ts = np.full(ncol, float(col["T_surface"]))
    Ts[0] *= 1.01
    for c in range(1, ncol):
        T[c - 1, c] *= 1.01
    fup, fdn, ftot = r.radiate_ir_batch(Ts, T)
    scale = np.max(np.abs(fup))
    for c in range(ncol):
        w = S.Column(col)
        w["T"] = T[:, c].copy()
        w["T_surface"] = Ts[c]
        r.radiate(*w.args(), compute_solar=False, compute_opacity=False)
        assert np.max(np.abs(fup[:, c] - np.array(r.wrk_ir.fup_n))) <= 1e-11 * scale, c
        assert np.max(np.abs(fdn[:, c] - np.array(r.wrk_ir.fdn_n))) <= 1e-11 * scale, c
        assert np.max(np.abs(ftot[:, c] - np.array(r.f_total))) <= 1e-11 * max(scale, np.max(np.abs(ftot))), c
    o = O.OracleRadtran(tb, nz, 4, 0.15)
    o.radiate(*col.args())
    for c in (0, 137, 402):
        w = S.Column(col)
        w["T"] = T[:, c].copy()
        w["T_surface"] = Ts[c]
        o.radiate(*w.args(), compute_solar=False, compute_opacity=False)
        assert np.max(np.abs(fup[:, c] - o.wrk_ir.fup_n)) <= 1e-9 * np.max(np.abs(o.wrk_ir.fup_n))
        assert np.max(np.abs(ftot[:, c] - o.f_total)) <= 1e-9 * np.max(np.abs(o.f_total))
