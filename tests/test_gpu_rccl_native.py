"""The library's OWN multi-GPU step (include/clima_radtran_hip.h, radtran_comm_*): shard -> kernels ->
ncclAllReduce on the handle's stream -> f_total, behind the C ABI and the Fortran shim -- no torch, no
torch.distributed.  A one-GPU box allows one rank per communicator; that still drives RCCL itself, the status
word that rides on the all-reduce, the partial-row bookkeeping of IR-only steps and the Fortran bindings.  The
shard arithmetic for N > 1 is exercised by giving one-rank communicators a rehearsed shard (rank r of N) and
adding the ranks' results on the host."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _levels(r):
    return np.concatenate([r.wrk_ir.fup_n, r.wrk_ir.fdn_n, r.wrk_sol.fup_n, r.wrk_sol.fdn_n])


def test_one_rank_communicator_is_bit_identical_to_the_plain_call(hip_lib, small_tables):
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    nz = 80
    col = S.modern_earth_column(nz)
    ref = Radtran(small_tables, nz, 4, 0.25)
    want = ref.TOA_fluxes(*col.args())
    want_lv, want_f = _levels(ref), np.array(ref.f_total)

    r = Radtran(small_tables, nz, 4, 0.25)
    assert r.comm() == (0, 0, 0)
    r.comm_init_rank(1, 0, Radtran.comm_unique_id())
    assert r.comm()[:2] == (1, 0)
    assert r.TOA_fluxes(*col.args()) == want                       # same kernels, a one-rank sum: same bits
    np.testing.assert_array_equal(_levels(r), want_lv)
    np.testing.assert_array_equal(np.array(r.f_total), want_f)
    assert r.comm()[2] == 1                                        # exactly one collective per step
    # resident form: three steps, no host round trip in between
    r.upload_column(*col.args())
    for _ in range(3):
        r.radiate_resident()
    r.synchronize()
    assert r.comm()[2] == 4
    np.testing.assert_array_equal(np.array(r.f_total), want_f)
    # the RCE-Jacobian pattern: IR only on stored opacities, warmer column; the solar rows must survive
    warm = S.Column(col)
    warm["T"] = col["T"] + 2.0
    warm["T_surface"] = col["T_surface"] + 2.0
    a = ref.TOA_fluxes(*warm.args(), compute_solar=False, compute_opacity=False)
    b = r.TOA_fluxes(*warm.args(), compute_solar=False, compute_opacity=False)
    assert a == b
    np.testing.assert_array_equal(_levels(r), _levels(ref))
    np.testing.assert_array_equal(np.array(r.f_total), np.array(ref.f_total))
    # the batched entry points refuse a sharded handle only when it IS sharded; leaving the communicator
    r.comm_destroy()
    assert r.comm()[0] == 0 and r.bin_shard()[1] == small_tables.nw
    assert r.TOA_fluxes(*col.args()) == want


def test_shards_of_one_rank_communicators_add_up(hip_lib, small_tables):
    """World 3 rehearsed on one GPU: three handles, each with its own one-rank communicator and the bin shard
    (k, 3).  Every step runs the library's all-reduce (over one rank: the partial rows come back unchanged), so
    the host-side sum of the three must be the whole spectrum's result -- including after an IR-only step, where
    each handle has to put its PARTIAL solar rows back before its next reduce."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    nz, W = 70, 3
    col = S.modern_earth_column(nz)
    ref = Radtran(small_tables, nz, 2, 0.2)
    ref.radiate(*col.args())
    want_lv = _levels(ref)
    parts = []
    for k in range(W):
        r = Radtran(small_tables, nz, 2, 0.2)
        r.comm_init_rank(1, 0, Radtran.comm_unique_id())
        r.set_bin_shard(k, W)                                      # allowed on a one-rank communicator only
        r.radiate(*col.args())
        parts.append(r)
    bins = [p.bin_shard() for p in parts]
    assert sum(b[1] for b in bins) == small_tables.nw and all(b[1] > 0 for b in bins)
    got = sum(_levels(p) for p in parts)
    np.testing.assert_allclose(got, want_lv, rtol=1e-13, atol=1e-13 * np.max(np.abs(want_lv)))
    warm = S.Column(col)
    warm["T"] = col["T"] + 1.5
    ref.radiate(*warm.args(), compute_solar=False, compute_opacity=False)
    for p in parts:
        p.radiate(*warm.args(), compute_solar=False, compute_opacity=False)
        p.radiate(*warm.args(), compute_solar=False, compute_opacity=False)   # twice: partial rows, not reduced ones, go back
    got = sum(_levels(p) for p in parts)
    want_lv = _levels(ref)
    np.testing.assert_allclose(got, want_lv, rtol=1e-13, atol=1e-13 * np.max(np.abs(want_lv)))


def test_a_multi_rank_shard_is_fixed_by_the_communicator(hip_lib, small_tables):
    from clima_amd.radtran import Radtran, ClimaException
    r = Radtran(small_tables, 40, 2, 0.2)
    with pytest.raises(ClimaException, match="invalid communicator"):
        r.comm_init_rank(2, 2, Radtran.comm_unique_id())
    r.comm_init_rank(1, 0, Radtran.comm_unique_id())
    with pytest.raises(ClimaException, match="already has a communicator"):
        r.comm_init_rank(1, 0, Radtran.comm_unique_id())


def test_handoff_timeout_on_a_communicator_handle_is_repeated_not_reported(hip_lib, monkeypatch):
    """With a communicator the partial rows are summed before the host can look, so an expired hand-off wait
    cannot be repaired rank by rank: the status word behind the level rows carries it to every rank and all of
    them repeat the step through the separate launches (round 2 returned an error here)."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    tables = S.modern_earth_tables(nw=400)
    nz = 200
    col = S.modern_earth_column(nz)
    ref = Radtran(tables, nz, 4, 0.2)
    ref.fused = False
    want = ref.TOA_fluxes(*col.args())
    want_f = np.array(ref.f_total)
    monkeypatch.setenv("CLIMA_HIP_FUSED_SPINS", "0")
    r = Radtran(tables, nz, 4, 0.2)
    monkeypatch.delenv("CLIMA_HIP_FUSED_SPINS")
    r.comm_init_rank(1, 0, Radtran.comm_unique_id())
    r.coop_items = 0                                               # keep the fused grid for this size
    assert r.fused
    got = r.TOA_fluxes(*col.args())
    assert r.fused_fallbacks >= 1
    assert got == want
    np.testing.assert_array_equal(np.array(r.f_total), want_f)
    n0 = r.fused_fallbacks
    r.upload_column(*col.args())
    r.radiate_resident()
    r.synchronize()                                                # resident form: detected at the synchronise
    assert r.fused_fallbacks > n0
    np.testing.assert_array_equal(np.array(r.f_total), want_f)


def test_fortran_driver_in_sharded_mode(hip_lib, tmp_path):
    """`radtran_driver case res <rank> <nranks> <id-file>`: the Fortran host joins a communicator through
    rad%comm_init_file and calls the SAME rad%radiate / rad%TOA_fluxes; one rank must reproduce the plain run
    bit for bit."""
    from clima_amd import build, synthetic as S
    from clima_amd.fortran_case import write_case
    build.build()
    exe = build.build_fortran_shim()
    if exe is None:
        pytest.skip("amdflang is not available on this box")
    tb = S.modern_earth_tables(nw=30)
    nz, nzen, albedo = 40, 4, 0.15
    col = S.modern_earth_column(nz)
    case = str(tmp_path / "case.bin")
    write_case(case, tb, col, nzen, albedo)
    plain, shard = str(tmp_path / "plain.txt"), str(tmp_path / "shard.txt")
    out = subprocess.run([exe, case, plain], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    idf = str(tmp_path / "comm.id")
    out = subprocess.run([exe, case, shard, "0", "1", idf, "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert not os.path.exists(idf)                                 # rank 0 removes the rendezvous file
    a = np.array(open(plain).read().split(), dtype=float)
    b = np.array(open(shard).read().split(), dtype=float)
    nw_ir, nw_sol = len(tb.ir_wavl) - 1, len(tb.sol_wavl) - 1
    n = 2 + 3 * (nz + 1) + nw_ir + nw_sol                          # ISR, OLR, three level rows, the two TOA spectra
    assert len(b) == n
    np.testing.assert_array_equal(a[:n], b)


def test_radiation_enhancement_on_a_communicator_handle(hip_lib, small_tables):
    """Radtran%apply_radiation_enhancement (clima_radtran.f90:402-411) on a handle whose level rows are reduced in
    place: the slot behind the rows is the step's status word there, not f_total -- the scaling must neither disturb it
    nor be undone by a repair that runs afterwards; IR-only steps that follow keep the enhanced solar rows."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    nz = 70
    col = S.modern_earth_column(nz)
    ref = Radtran(small_tables, nz, 2, 0.2)
    ref.radiate(*col.args())
    ref.apply_radiation_enhancement(1.6)
    want_sol, want_f = np.array(ref.wrk_sol.fdn_n), np.array(ref.f_total)
    parts = []
    for k in range(2):
        r = Radtran(small_tables, nz, 2, 0.2)
        r.comm_init_rank(1, 0, Radtran.comm_unique_id())
        r.set_bin_shard(k, 2)
        r.upload_column(*col.args())
        r.radiate_resident()                       # not synchronised: the enhancement settles the step itself
        r.apply_radiation_enhancement(1.6)
        assert r.fused_fallbacks == 0
        parts.append(r)
    got_sol = sum(np.array(p.wrk_sol.fdn_n) for p in parts)
    np.testing.assert_allclose(got_sol, want_sol, rtol=1e-13)
    got_f = sum(np.array(p.f_total) for p in parts)
    np.testing.assert_allclose(got_f, want_f, rtol=1e-12, atol=1e-12 * np.max(np.abs(want_f)))
    warm = S.Column(col)
    warm["T"] = col["T"] + 1.0
    ref.radiate(*warm.args(), compute_solar=False, compute_opacity=False)
    for p in parts:
        p.radiate(*warm.args(), compute_solar=False, compute_opacity=False)
    got_sol = sum(np.array(p.wrk_sol.fdn_n) for p in parts)
    np.testing.assert_allclose(got_sol, want_sol, rtol=1e-13)      # the enhanced partial rows were put back
    got_f = sum(np.array(p.f_total) for p in parts)
    np.testing.assert_allclose(got_f, np.array(ref.f_total), rtol=1e-12, atol=1e-12 * np.max(np.abs(want_f)))


@pytest.mark.parametrize("mode", [0, 2])
def test_ir_batch_on_communicator_handles(hip_lib, small_tables, mode):
    """radtran_radiate_ir_batch (the RCE Jacobian's batch) on a handle with a communicator: the rank's share of the bins
    -> ONE all-reduce of the batch's up / down arrays -> f_total from the reduced rows.  One rank: bit for bit the plain
    call; three rehearsed shards: their rows add up to it.  General kernel (mode 0) and response form (mode 2)."""
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran
    nz, W, ncol = 60, 3, 14
    col = S.modern_earth_column(nz)
    T = np.repeat(np.asarray(col["T"], dtype=float)[:, None], ncol, axis=1)
    Ts = np.full(ncol, float(col["T_surface"]))
    Ts[0] += 1.0
    for c in range(1, ncol):
        T[(5 * c) % nz, c] += 0.5 + 0.1 * c
    T[:, 3] += np.linspace(0.0, 2.0, nz)                           # a dense column
    ref = Radtran(small_tables, nz, 2, 0.2)
    ref.ir_green = mode
    ref.radiate(*col.args())
    want = ref.radiate_ir_batch(Ts, T)

    one = Radtran(small_tables, nz, 2, 0.2)
    one.ir_green = mode
    one.comm_init_rank(1, 0, Radtran.comm_unique_id())
    one.radiate(*col.args())
    n0 = one.comm()[2]
    got = one.radiate_ir_batch(Ts, T)
    assert one.comm()[2] == n0 + 1                                 # exactly one collective per batch
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(np.array(one.f_total), np.array(ref.f_total))   # the handle's own state is untouched

    parts = []
    for k in range(W):
        r = Radtran(small_tables, nz, 2, 0.2)
        r.ir_green = mode
        r.comm_init_rank(1, 0, Radtran.comm_unique_id())
        r.set_bin_shard(k, W)
        r.radiate(*col.args())
        parts.append(r.radiate_ir_batch(Ts, T))
    for i in range(3):
        s = sum(p[i] for p in parts)
        np.testing.assert_allclose(s, want[i], rtol=1e-12, atol=1e-12 * np.max(np.abs(want[i])))
    plain = Radtran(small_tables, nz, 2, 0.2)
    plain.set_bin_shard(0, 2)                                       # a shard without a communicator: nobody would reduce
    plain.radiate(*col.args())
    from clima_amd.radtran import ClimaException
    with pytest.raises(ClimaException, match="bin-sharded"):
        plain.radiate_ir_batch(Ts, T)


def test_comm_init_file_refuses_another_jobs_record(tmp_path, monkeypatch):
    """radtran_comm_init_file, a rank other than 0: a record at the path whose nonce (CLIMA_COMM_NONCE) or communicator size
    is not this job's -- the leftover of a crashed or re-launched job -- is not joined (ncclCommInitRank on a dead id would
    block for ever): the call returns an error once its wait is over."""
    import struct
    from clima_amd import synthetic as S
    from clima_amd.radtran import ClimaException, Radtran
    r = Radtran(S.modern_earth_tables(nw=20), 30, 2, 0.2)
    path = str(tmp_path / "id.bin")
    # a well-formed record of ANOTHER job: magic, nranks = 2, nonce "old-job", 128 id bytes
    with open(path, "wb") as f:
        f.write(b"CLRCOMM1" + struct.pack("<i", 2) + b"old-job".ljust(64, b"\0") + bytes(128))
    monkeypatch.setenv("CLIMA_COMM_NONCE", "this-job")
    monkeypatch.setenv("CLIMA_COMM_WAIT_S", "1")
    with pytest.raises(ClimaException, match="a record of another job"):
        r.comm_init_file(2, 1, path)
    assert r.comm() == (0, 0, 0) or r.comm()[0] == 0
