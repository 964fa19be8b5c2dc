#!/usr/bin/env python3
"""Benchmark of the radiate() hot path (BASELINE.json: `radiate() calls/sec`, 200-layer
ModernEarth column, full solar+IR correlated-k bin grid).

  python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

A "step" is ONE full `Radtran%radiate` call -- opacity assembly + IR + solar two-stream +
spectral integration -- on the ModernEarth column of tests/test_radtran.f90 (config 2:
nz=200, nw=1000 opacity bins (600 IR / 600 solar), 8 g-points, 8 zenith angles, 5
k-distribution species, synthetic tables: SURVEY.md 8(d)).  Inputs (tables and the
column) are resident in HBM before the timed region starts.

What the one JSON line carries (rank 0):
  value          = `pipelined_calls_per_s`: the driver's contract -- K calls enqueued back to back on the
                 handle's stream, bracketed by barrier + synchronise, inputs resident; the MEDIAN of at
                 least `--repeats` such K-step timings, and of as many as it takes to time ~0.3 s in all
                 (a 20-step run is 2.5 ms: one timing is a sample of the clock ramp); their spread is in
                 `repeats_calls_per_s`.  It is NOT the per-call latency of SURVEY.md 8(d) "Metric"; those are:
  resident_sync  one `radiate_resident` + `synchronize` per call (column in HBM, one host round trip per
                 call), timed call by call inside the library; median, p10, p90 over 200 calls.
  sync_api_c     the drop-in call of SURVEY.md 8(d) "Metric": `radtran_toa_fluxes_wrapper` with host
                 arrays in, ISR / OLR out, one stream synchronise per call (PCIe inclusive), timed call by
                 call INSIDE the library (what a Fortran / C caller of `TOA_fluxes` sees, and what
                 `cpu_baseline` is comparable with); median, p10, p90 over 200 calls.
  sync_api       the same call through the Python ctypes mirror (adds the foreign-function layer's ~14 us).
  roofline       dominant kernel, HIP events on the library's stream inside the timed region;
                 `fp64_issue_frac` = VALU instructions per launch (PMC pass under profiles/) x 4
                 cycles / (1024 SIMDs x 2.4 GHz) / kernel time; `traffic` from the PMC pass.  Both PMC
                 figures are printed only when clima_amd/csrc/kernels.hip still has the hash recorded
                 beside the PMC summary (profiles/r04_pmc.json) -- otherwise null.
  algorithmic    N_PT, N_T and both byte variants of SURVEY.md 8(d).
  cpu_baseline   the oracle on this box's host cores, same workload (a reported baseline).

N > 1 (launched by torch.distributed.run, one rank per GPU): the spectral bins of the SAME call are
sharded over the ranks (work-balanced contiguous ranges) and every step ends with one RCCL
all-reduce of the 4*(nz+1) partial level fluxes -- strong scaling of one call.  The collective is the
LIBRARY's own (radtran_comm_init_rank: ncclAllReduce on the handle's stream inside radiate_resident);
torch.distributed only launches the ranks, hands the communicator id round and takes the maximum of the
ranks' timings.  CLIMA_BENCH_TORCH_ALLREDUCE=1 selects round 2's form (torch's all_reduce on a tensor
aliasing the library's buffer) as a cross-check.

--config 3 (EarlyMars, CIA-heavy, 200 layers), 4 (1024 perturbed columns through
radtran_toa_fluxes_batch: columns/s; with N > 1 every rank takes 1024/N columns, no collective) and
5 (one 500-layer column; with N > 1 bins sharded + all-reduce) print the same kind of line for the
other BASELINE.json configurations; the driver's default run is config 2.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS, CLOCK_GHZ, F64_CYCLES = 1024, 2.4, 4.0   # 256 CUs x 4 SIMDs; one wave64 f64 instruction per SIMD per 4 cycles
CLOCK_MEASURED_GHZ = 1.97   # what the part holds under this load (s_memtime against s_memrealtime, profiles/r03_stamps.txt)
KERNELS = ["prep", "opacity", "twostream", "integrate"]
EVENT_STRIDE = 16  # HIP events around the dominant kernel on every 16th launch of the timed region (first one included)
PMC_FILE = os.path.join(ROOT, "profiles", "r04_pmc.json")
KERNEL_SRC = os.path.join(ROOT, "clima_amd", "csrc", "kernels.hip")


def pmc_for(kernel, workload_key):
    """PMC figures of `kernel` on `workload_key` from the committed pass -- only if the kernels are
    still the ones that were profiled."""
    try:
        with open(PMC_FILE) as f:
            rec = json.load(f)
        with open(KERNEL_SRC, "rb") as f:
            h = hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None, "no PMC record (%s)" % os.path.relpath(PMC_FILE, ROOT)
    if rec.get("kernels_hip_sha256") != h:
        return None, "stale: kernels.hip changed since the PMC pass of %s" % rec.get("source")
    k = rec.get("workloads", {}).get(workload_key, {}).get(kernel)
    if not k:
        return None, "no PMC pass for %s on %s" % (kernel, workload_key)
    return k, rec.get("source")


def visible_gpus():
    """Number of devices this process could open, WITHOUT initialising the GPU (on this image
    torch.cuda.device_count() reads the topology only)."""
    import torch
    return int(torch.cuda.device_count())


def launcher_command(n, argv, port=None):
    """The command that starts the N ranks: the driver's own form of the N > 1 launch."""
    if port is None:
        port = 29500 + (os.getpid() % 400)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv, run=None, ndev=None):
    """`python bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): start the N ranks as fresh
    child processes -- this process has not touched the GPU and never will -- pass rank 0's JSON line through,
    exit with the children's code.  Before the run proper the library's own RCCL step (communicator created by
    radtran_comm_init_rank, ncclAllReduce on the handle's stream) is tried on the same N ranks in a short child job
    under a time limit: it has not run on more than one GPU yet, so the measured run takes it only when
    that job came back clean and otherwise keeps torch.distributed's all-reduce."""
    import subprocess
    run = run or subprocess.run
    ndev = visible_gpus() if ndev is None else ndev
    if ndev < n:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (n, ndev))
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    if "CLIMA_BENCH_NATIVE_ALLREDUCE" not in env and "CLIMA_BENCH_TORCH_ALLREDUCE" not in env:
        probe_env = dict(env, CLIMA_BENCH_NATIVE_ALLREDUCE="1")
        try:
            pr = run(launcher_command(n, ["--gpus", str(n), "--probe-native"]), env=probe_env, timeout=240,
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            ok = pr.returncode == 0 and b"native step ok" in (pr.stdout or b"")
        except subprocess.TimeoutExpired:
            ok = False
        sys.stderr.write("bench.py: library-owned RCCL step on %d ranks: %s\n" % (n, "ok, used" if ok else "NOT clean, torch's all-reduce used"))
        env["CLIMA_BENCH_NATIVE_ALLREDUCE"] = "1" if ok else "0"
    r = run(launcher_command(n, argv), env=env)
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--probe-native", action="store_true",
                    help="(launcher's use) three steps through the library's own RCCL step on the N ranks, checked against "
                         "the same steps with torch's all-reduce; prints 'native step ok'")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--nz", type=int, default=None)
    ap.add_argument("--nzen", type=int, default=None)
    ap.add_argument("--ncol", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-jacobian", action="store_true", help="skip the RCE-Jacobian batch figure (rce_jacobian_batch)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # stdout carries exactly one JSON line: anything libraries print there (RCCL's version
    # banner at communicator creation, for one) is routed to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d inside a job of %d rank(s)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Radtran hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # CLIMA_BENCH_FORCE_DIST=1 runs the N>1 step (shard -> RCCL all-reduce -> finish) with one
    # rank, to rehearse that code path on a one-GPU box
    dist_on = world > 1 or os.environ.get("CLIMA_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran

    cfg = args.config
    if cfg == 3:      # EarlyMars, CIA-heavy, nzen 4, albedo 0.2, photon scale 0.4286
        nz, nzen, albedo = args.nz or 200, args.nzen or 4, 0.2
        tables, col = S.early_mars_tables(), S.early_mars_column(nz)
        what = "EarlyMars column (CO2-dominated, CIA-heavy), one Radtran%%radiate call: nz=%d, nw=1000, 8 g-points, %d zenith angles; config 3 of BASELINE.json" % (nz, nzen)
    elif cfg == 5:    # one 500-layer column
        nz, nzen, albedo = args.nz or 500, args.nzen or 8, 0.15
        tables, col = S.modern_earth_tables(), S.modern_earth_column(nz)
        what = "ModernEarth column, one Radtran%%radiate call: nz=%d, nw=1000, 8 g-points, %d zenith angles; config 5 of BASELINE.json" % (nz, nzen)
    else:
        nz, nzen, albedo = args.nz or 200, args.nzen or 8, 0.15
        tables, col = S.modern_earth_tables(), S.modern_earth_column(nz)
        what = ("ModernEarth column, one Radtran%%radiate call: nz=%d, nw=1000 (600 IR + 600 solar bins), 8 g-points, "
                "%d zenith angles, nk=5; config 2 of BASELINE.json" % (nz, nzen))
    rad = Radtran(tables, nz, nzen, albedo)  # tests/test_radtran.f90:35-38
    if cfg == 3:
        rad.photon_scale_factor = 0.4286

    if cfg == 4:
        return config4(args, rad, tables, nz, nzen, world, rank, dist_on, dist, torch, json_fd)

    # Which all-reduce ends a sharded step.  The library's own (radtran_comm_init_rank: ncclAllReduce on the handle's
    # stream) is what a Fortran / C host gets and what the one-rank rehearsals and tests run; it has not yet run
    # across two GPUs (no box of this pool offers two), so for world > 1 it is taken only when asked for:
    # CLIMA_BENCH_NATIVE_ALLREDUCE=1 -- which `python bench.py --gpus N` sets by itself after trying that step on the
    # N ranks in a short child job (self_launch) -- and torch.distributed's all-reduce on a tensor aliasing the
    # library's buffer otherwise.  CLIMA_BENCH_TORCH_ALLREDUCE=1 forces the latter also for the rehearsal.
    if os.environ.get("CLIMA_BENCH_TORCH_ALLREDUCE") == "1":
        torch_ar = True
    elif world > 1:
        torch_ar = os.environ.get("CLIMA_BENCH_NATIVE_ALLREDUCE") != "1"
    else:
        torch_ar = False
    if args.probe_native:
        return probe_native(rad, col, nz, world, rank, dist, torch, json_fd)
    if dist_on:
        if torch_ar:
            rad.set_bin_shard(rank, world)
        else:
            # the library's own step: rank 0 draws the communicator id, torch.distributed hands it round
            ids = [Radtran.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            rad.comm_init_rank(world, rank, ids[0])
        fake = os.environ.get("CLIMA_BENCH_FAKE_SHARD")   # "rank,world": rehearse one rank's share of an N-GPU step on one GPU
        if fake and world == 1:
            rad.set_bin_shard(*[int(x) for x in fake.split(",")])
    rad.upload_column(*col.args())
    flux = rad.flux_tensor() if (dist_on and torch_ar) else None
    # The all-reduce is ordered against the library's kernels on the device: the library's HIP
    # stream is made torch's current stream for the collective, so RCCL's stream waits for the
    # partial fluxes and the library stream waits for the reduced ones -- no host round trip
    # inside a step.  CLIMA_BENCH_HOST_SYNC=1 selects the plain host-synchronised form.
    host_sync = os.environ.get("CLIMA_BENCH_HOST_SYNC") == "1"
    lib_stream = None
    if dist_on and torch_ar and not host_sync:
        try:
            lib_stream = torch.cuda.ExternalStream(rad.stream())
        except Exception as e:  # same torch build on every rank: all of them take the same branch
            print("bench: torch.cuda.ExternalStream unavailable (%s); host-synchronised steps" % e, file=sys.stderr)
            host_sync = True

    def step():
        rad.radiate_resident()                    # with a communicator: shard -> kernels -> ncclAllReduce, all on the library's stream
        if dist_on and torch_ar:
            if host_sync:
                rad.synchronize()                 # library stream -> host
                dist.all_reduce(flux)             # RCCL over xGMI: 4*(nz+1) doubles
                torch.cuda.current_stream().synchronize()
            else:
                with torch.cuda.stream(lib_stream):
                    dist.all_reduce(flux)
            rad.finish_reduced()              # f_total from the reduced fluxes

    def barrier():
        # the library's stream first: its own collectives (a communicator of its own) have drained on every rank
        # before torch's barrier runs, so kernels of the two communicators never wait on each other on one GPU
        rad.synchronize()
        if dist_on:
            dist.barrier()
        rad.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # HIP events bracket the dominant kernel on the library's stream during the timed region, on
    # every EVENT_STRIDE-th launch (an event pair drains the queue for ~5 us; sampled, the
    # measurement costs the measured throughput ~0.3 us per step instead): the average of these
    # durations is roofline.achieved's denominator.
    # The short fully instrumented pass (per-kernel breakdown) runs first, so that the timed
    # region is not also the one in which the device clocks settle.
    rad.profile_stride(1)
    rad.profile(True)
    rad.profile_reset()
    for _ in range(min(max(args.steps, 1), 50)):
        step()
    barrier()
    kt = [rad.kernel_time(i) for i in range(4)]
    kt_all = list(kt)          # the fully instrumented pass, kept apart from the timed region's own sample of the dominant kernel
    rad.profile(2)
    rad.profile_stride(EVENT_STRIDE)
    rad.profile_reset()
    dts = []
    repeats = max(args.repeats, 1)
    while len(dts) < repeats:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        dts.append(dt)
        if len(dts) == 1:
            # a short K-step loop (the driver's 20 steps are 2.5 ms) is repeated until ~0.3 s have been timed
            # in all, so that the median is taken at settled clocks (at most 200 repeats; with N ranks dt is
            # the all-reduced maximum, the same number on every rank, so all of them loop alike)
            repeats = max(repeats, min(200, int(0.3 / max(dt, 1e-6)) + 1))
    dt = float(np.median(dts))
    kt_dom = rad.kernel_time(1)
    if kt_dom[1] > 0:
        kt[1] = kt_dom
    rad.profile_stride(1)
    rad.profile(False)

    # ---- per-call latencies (SURVEY.md 8(d) "Metric"): one host round trip per call
    def _stats(ts, what):
        ts = np.asarray(ts, dtype=float)
        return {"what": what, "calls_per_s": 1e6 / float(np.median(ts)), "us_median": float(np.median(ts)),
                "us_p10": float(np.percentile(ts, 10)), "us_p90": float(np.percentile(ts, 90)), "n": int(len(ts))}

    sync_api = sync_api_c = resident_sync = None
    a = col.args()
    rad.bench_resident_sync(20)
    resident_sync = _stats(rad.bench_resident_sync(200),
                           "radtran_radiate_resident + radtran_synchronize per call (column resident in HBM, one host round trip "
                           "per call%s), timed call by call inside the library" % (", all-reduce included" if dist_on else ""))
    if not dist_on:
        rad.bench_toa_fluxes(20, *a)
        sync_api_c = _stats(rad.bench_toa_fluxes(200, *a),
                            "radtran_toa_fluxes_wrapper: host arrays in, ISR/OLR out, one stream synchronise per call (PCIe "
                            "inclusive), timed call by call inside the library (what a Fortran / C host sees)")
        for _ in range(10):
            rad.TOA_fluxes(*a)
        ts = []
        for _ in range(100):
            t0 = time.perf_counter()
            rad.TOA_fluxes(*a)
            ts.append(time.perf_counter() - t0)
        sync_api = _stats(np.array(ts) * 1e6, "the same call through the ctypes mirror clima_amd.radtran.Radtran.TOA_fluxes")
        rad.upload_column(*a)

    # ---- companion figure (not `value`): the RCE Jacobian's batch on this column's grid -- nz+1 IR-only
    # calls on the resident opacities, each with one temperature changed, through radtran_radiate_ir_batch: the general
    # batch kernel and the response form the library takes for such a batch by itself
    jac = None
    if not dist_on and cfg in (2, 3, 5) and not args.no_jacobian:
        ncj = nz + 1
        Tj = np.repeat(np.asarray(col["T"], dtype=float)[:, None], ncj, axis=1)
        Tsj = np.full(ncj, float(col["T_surface"]))
        Tsj[0] += 1.0
        for c in range(1, ncj):
            Tj[c - 1, c] += 1.0
        jres = {}
        for mode in (0, 1):
            rad.ir_green = mode
            n0 = rad.ir_green_batches
            outj = rad.radiate_ir_batch(Tsj, Tj)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                rad.radiate_ir_batch(Tsj, Tj, out=outj)
                best = min(best, time.perf_counter() - t0)
            jres[mode] = (best, [np.array(x) for x in outj], rad.ir_green_batches > n0)
        rad.ir_green = 1
        # the same batch with the caller's three result arrays page-locked (radtran_batch_pin_results_set: a caller that keeps them)
        bestp = 1e9
        for k in range(4):
            t0 = time.perf_counter()
            rad.radiate_ir_batch(Tsj, Tj, out=outj, pin=True)
            if k:
                bestp = min(bestp, time.perf_counter() - t0)
        rad.spectra_release()
        dj = max(float(np.max(np.abs(x - y)) / np.max(np.abs(y))) for x, y in zip(jres[1][1], jres[0][1]))
        jac = {"what": "radtran_radiate_ir_batch: %d IR-only columns x %d layers on the resident opacities, one temperature changed "
                       "per column (src/adiabat/clima_adiabat_solve.f90:798-812), host arrays in and out, best of 3" % (ncj, nz),
               "ms": 1e3 * jres[1][0], "us_per_column": 1e6 * jres[1][0] / ncj, "response_form": bool(jres[1][2]),
               "ms_result_arrays_page_locked": 1e3 * bestp,
               "general_kernel_ms": 1e3 * jres[0][0], "largest_difference_of_row_maximum": dj}
        rad.upload_column(*a)

    # ---- companion figure for N > 1 (not `value`): the column-parallel form of config 4 --
    # every rank runs whole, unsharded calls on its own column, no collective (weak scaling)
    col_par = None
    if dist_on:
        rad2 = Radtran(tables, nz, nzen, albedo)
        rad2.upload_column(*col.args())
        for _ in range(args.warmup):
            rad2.radiate_resident()
        dist.barrier(); rad2.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            rad2.radiate_resident()
        rad2.synchronize(); dist.barrier()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        col_par = {"value": world * args.steps / float(t[0]), "unit": "calls/s", "scaling": "weak",
                   "what": "each rank runs its own whole radiate() calls (independent columns), no collective"}
        del rad2

    # ---- parity of what was just timed (rank 0 checks OLR against the oracle)
    if dist_on and torch_ar:
        isr = float((flux[3 * (nz + 1) + nz] - flux[2 * (nz + 1) + nz]).item())
        olr = -float((flux[1 * (nz + 1) + nz] - flux[0 * (nz + 1) + nz]).item())
    else:
        step()
        rad.synchronize()
        w_ir, w_sol = rad.wrk_ir, rad.wrk_sol    # with a communicator: the reduced rows, the same on every rank
        olr = -(w_ir.fdn_n[nz] - w_ir.fup_n[nz])
        isr = w_sol.fdn_n[nz] - w_sol.fup_n[nz]

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = args.steps / dt
        names = list(KERNELS)
        if rad.fused and kt[2][1] == 0:   # opacity and two-stream work ran as one grid (k_fused), timed as kernel id 1
            names[1] = "fused"
        per_kernel_us = {k: (1e3 * ms / n if n else 0.0) for k, (ms, n) in zip(names, kt) if n}
        per_kernel_us_all = {k: (1e3 * ms / n if n else 0.0) for k, (ms, n) in zip(names, kt_all) if n}
        dom = max(per_kernel_us, key=per_kernel_us.get)
        ab = rad.algorithmic_bytes()
        nodes = rad.algorithmic_nodes()
        # algorithmic bytes of one call: distinct table nodes + inputs + outputs,
        # prorated to the bins this rank owns
        frac = rad.bin_shard()[1] / float(tables.nw)
        b_alg = (ab["tables_distinct"] + ab["output"]) * frac + ab["input"]
        b_alg_full = (ab["tables_full"] + ab["output"]) * frac + ab["input"]
        dur = per_kernel_us[dom] * 1e-6
        wkey = "config%d_nz%d_nzen%d" % (cfg, nz, nzen)
        pmc, pmc_src = (None, "bins sharded: no PMC pass") if dist_on else pmc_for("k_" + dom, wkey)
        issue = None
        if pmc and dur > 0 and pmc.get("SQ_INSTS_VALU"):
            issue = pmc["SQ_INSTS_VALU"] * F64_CYCLES / (SIMDS * CLOCK_GHZ * 1e9) / dur
        # what actually bounds the kernel: wave64 f64 issue.  At one instruction per SIMD per
        # 4 cycles the kernel's VALU instructions alone take valu*4/(1024*2.4e9) s: that is the floor of THIS
        # instruction stream, and the fraction of the HBM roof it would reach
        practical = None
        if pmc and pmc.get("SQ_INSTS_VALU"):
            floor_s = pmc["SQ_INSTS_VALU"] * F64_CYCLES / (SIMDS * CLOCK_GHZ * 1e9)
            floor_m = pmc["SQ_INSTS_VALU"] * F64_CYCLES / (SIMDS * CLOCK_MEASURED_GHZ * 1e9)
            practical = {"floor_us": floor_s * 1e6, "hbm_frac_at_floor": (b_alg / floor_s / 1e9) / HBM_PEAK_GBS,
                         "floor_us_at_measured_clock": floor_m * 1e6, "measured_clock_ghz": CLOCK_MEASURED_GHZ,
                         "hbm_frac_at_floor_measured_clock": (b_alg / floor_m / 1e9) / HBM_PEAK_GBS,
                         "what": "the kernel's %.4g VALU instructions per launch at one wave64 f64 instruction per SIMD per 4 cycles "
                                 "(1024 SIMDs) at the contract's 2.4 GHz and at the %.2f GHz the part holds under this load: no launch "
                                 "of this instruction stream can be shorter, so the 40 %% HBM target of north_star is out of reach "
                                 "for this algorithm" % (pmc["SQ_INSTS_VALU"], CLOCK_MEASURED_GHZ)}
        roofline = {"bound": "hbm", "practical_bound": "fp64_valu", "practical_ceiling": practical,
                    "kernel": "k_" + dom, "achieved": b_alg / dur / 1e9 if dur > 0 else 0.0,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (b_alg / dur / 1e9) / HBM_PEAK_GBS if dur > 0 else 0.0,
                    "traffic": (pmc["FETCH_SIZE_KiB"] + pmc["WRITE_SIZE_KiB"]) * 1024.0 if pmc else None,
                    "traffic_source": "FETCH_SIZE + WRITE_SIZE (KiB) of the kernel per launch, %s" % pmc_src,
                    "fp64_issue_frac": issue,
                    "fp64_issue_source": ("SQ_INSTS_VALU = %.4g per launch (%s) x %g cycles / (%d SIMDs x %g GHz) / kernel time"
                                          % (pmc["SQ_INSTS_VALU"], pmc_src, F64_CYCLES, SIMDS, CLOCK_GHZ)) if issue else pmc_src,
                    "algorithmic_bytes": b_alg,
                    "kernel_us": {dom: per_kernel_us[dom]},
                    "kernel_us_source": "HIP events around %s on every %d-th launch of the TIMED region (what `achieved` divides by)" % (dom, EVENT_STRIDE),
                    "kernel_us_instrumented_pass": per_kernel_us_all,
                    "kernel_us_instrumented_pass_source": "a separate pass of %d calls with an event pair around EVERY kernel, before the "
                                                          "timed region: each pair drains the queue, so these figures sit a few us above "
                                                          "the profiler's and their sum above ms_per_step -- a breakdown, not a budget" % min(max(args.steps, 1), 50),
                    "whole_call_frac": (b_alg / (dt / args.steps) / 1e9) / HBM_PEAK_GBS}
        out = {"metric": "radiate() calls/sec", "value": value, "unit": "calls/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": what,
                          "parallelism": ("bins sharded over %d GPUs + 1 all-reduce of %d f64" % (world, 4 * (nz + 1)))
                          if world > 1 else "1 GPU"},
               "value_is": "pipelined_calls_per_s: median of %d repeats of the %d-step loop, resident inputs, calls enqueued back to "
                           "back, one synchronise per loop (the driver's contract); per-call latencies are in resident_sync / sync_api_c" % (len(dts), args.steps),
               "pipelined_calls_per_s": value,
               "collective": (("library: ncclAllReduce on the handle's stream (radtran_comm_init_rank)" if not torch_ar else
                               "torch.distributed all_reduce on an aliased tensor") if dist_on else None),
               "repeats_calls_per_s": {"n": len(dts), "min": args.steps / max(dts), "p10": args.steps / float(np.percentile(dts, 90)),
                                       "median": args.steps / dt, "p90": args.steps / float(np.percentile(dts, 10)),
                                       "max": args.steps / min(dts), "first": args.steps / dts[0]},
               "olr_W_m2": olr / 1e3, "isr_W_m2": isr / 1e3, "roofline": roofline,
               "algorithmic": {"N_PT": nodes["N_PT"], "N_PT_table": nodes["N_PT_full"], "N_T": nodes["N_T"],
                               "N_T_table": nodes["N_T_full"], "bytes_distinct_nodes": b_alg,
                               "bytes_whole_tables": b_alg_full,
                               "frac_whole_tables": (b_alg_full / dur / 1e9) / HBM_PEAK_GBS if dur > 0 else 0.0}}
        if resident_sync is not None:
            out["resident_sync"] = resident_sync
        if sync_api_c is not None:
            out["sync_api_c"] = sync_api_c
        if sync_api is not None:
            out["sync_api"] = sync_api
        if col_par is not None:
            out["column_parallel"] = col_par
        if jac is not None:
            out["rce_jacobian_batch"] = jac
        if not args.no_cpu_baseline:   # rank 0, every N (the other ranks wait at the final barrier)
            out.update(cpu_baseline(tables, col, nz, nzen, albedo, olr, rad if world == 1 else None, 0.4286 if cfg == 3 else None))
            if jac is not None:   # the same batch as the reference issues it: nz+1 IR-only calls, here the oracle's on the host cores
                out["rce_jacobian_batch"]["cpu_port_ms"] = out["cpu_baseline"]["ir_only_call_ms"] * (nz + 1)
                out["rce_jacobian_batch"]["cpu_port_what"] = ("%d x one IR-only oracle call on the stored opacities (%d threads, %.1f ms each, "
                                                              "4 timed)" % (nz + 1, out["cpu_baseline"]["cores"], out["cpu_baseline"]["ir_only_call_ms"]))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist_on:
        dist.barrier()
        rad.comm_destroy()
        dist.destroy_process_group()


def probe_native(rad, col, nz, world, rank, dist, torch, json_fd):
    """The library-owned RCCL step on the job's ranks, checked against torch's all-reduce of the same partial rows:
    a full step and an IR-only step on the stored opacities (the partial solar rows are put back before the reduce)
    on the benchmark's own handle, then -- on a smaller spectrum (400 bins, 4 zenith angles: there two-stream blocks of
    the fused grid do find their tiles unfinished) -- a step with every hand-off wait of rank 0 made to expire
    (radtran_fused_spins_set(0)): the status word that rides on the all-reduce makes every rank repeat the step.
    Rank 0 prints `native step ok` when every row agrees to 1e-13 of its maximum on every rank."""
    import numpy as np
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran

    def rows(r):
        return np.concatenate([np.asarray(r.wrk_ir.fup_n), np.asarray(r.wrk_ir.fdn_n), np.asarray(r.wrk_sol.fup_n),
                               np.asarray(r.wrk_sol.fdn_n), np.asarray(r.f_total)])

    def by_torch(r, steps):
        """shard + torch all-reduce, host-synchronised: the plainest form"""
        r.set_bin_shard(rank, world)
        r.upload_column(*col.args())
        flux = r.flux_tensor()
        out = []
        for solar in steps:
            r.radiate_resident(compute_solar=solar)
            r.synchronize()
            dist.all_reduce(flux)
            torch.cuda.synchronize()
            r.finish_reduced()
            out.append(rows(r))
        r.set_bin_shard(0, 1)
        return out

    def attach(r):
        ids = [Radtran.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        r.comm_init_rank(world, rank, ids[0])
        r.upload_column(*col.args())

    want = by_torch(rad, (True, False))
    attach(rad)
    got = []
    for solar in (True, False):
        rad.radiate_resident(compute_solar=solar)
        rad.synchronize()
        got.append(rows(rad))
    rad.synchronize()
    dist.barrier()
    rad.comm_destroy()
    # the forced repeat
    small = Radtran(S.modern_earth_tables(nw=400), nz, 4, 0.2)
    want += by_torch(small, (True,))
    attach(small)
    n0 = small.fused_fallbacks
    if rank == 0:
        small.fused_spins = 0
    small.radiate_resident()
    small.synchronize()
    got.append(rows(small))
    repeats = small.fused_fallbacks - n0
    small.synchronize()
    dist.barrier()
    small.comm_destroy()
    err = max(float(np.max(np.abs(g - w)) / np.max(np.abs(w))) for g, w in zip(got, want))
    t = torch.tensor([err, float(repeats)], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    err, repeats = float(t[0]), int(t[1])
    dist.destroy_process_group()
    if rank == 0:
        msg = "native step %s: largest difference from torch's all-reduce %.2e of a row's maximum on %d ranks, repeated steps %d\n" % (
            "ok" if err < 1e-13 else "DIFFERS", err, world, repeats)
        os.write(json_fd, msg.encode())
    if err >= 1e-13:
        sys.exit(3)


def config4(args, rad, tables, nz, nzen, world, rank, dist_on, dist, torch, json_fd):
    """BASELINE.json config 4: `--ncol` perturbed ModernEarth columns (seed 7) through
    radtran_toa_fluxes_batch; a step = the whole batch; with N ranks every rank takes ncol/N columns,
    no collective (weak scaling in columns per rank is not what is asked: the batch is fixed -> strong)."""
    import numpy as np
    from clima_amd import synthetic as S
    cols = S.perturbed_columns(args.ncol, nz=nz, seed=7)
    mine = cols[rank::world] if world > 1 else cols
    steps, warm = max(1, min(args.steps, 5)), max(1, min(args.warmup, 2))
    for _ in range(warm):
        isr, olr = rad.TOA_fluxes_batch(mine)
    dts = []
    for _ in range(steps):
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        isr, olr = rad.TOA_fluxes_batch(mine)
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        dts.append(dt)
    dt = float(np.median(dts))
    if rank == 0:
        out = {"metric": "columns/sec (radtran_toa_fluxes_batch)", "value": args.ncol / dt, "unit": "columns/s",
               "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * dt, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%d perturbed ModernEarth columns (T-P and mixing-ratio sweep, seed 7), nz=%d, nw=1000, 8 g-points, "
                                      "%d zenith angles, host arrays in / ISR, OLR out per batch; config 4 of BASELINE.json" % (args.ncol, nz, nzen),
                          "parallelism": "%d GPU(s), columns split, no collective" % world},
               "us_per_column": 1e6 * dt / args.ncol * world, "value_is": "median of %d batches" % steps,
               "olr_W_m2_mean": float(np.mean(olr)) / 1e3, "fused_fallbacks": rad.fused_fallbacks}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist_on:
        dist.destroy_process_group()


def physical_cores():
    """(physical cores this process may use, how that was determined).  The CPUs of the affinity mask, counted
    once per (package, core) pair of /sys/devices/system/cpu/cpuN/topology -- SMT siblings share a pair -- and
    capped by the cgroup's CPU quota when there is one."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    seen, how = set(), "topology"
    for c in cpus:
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            with open(base + "physical_package_id") as f:
                pkg = f.read().strip()
            with open(base + "core_id") as f:
                core = f.read().strip()
            seen.add((pkg, core))
        except OSError:
            seen, how = set(), "no topology files"
            break
    n = len(seen) if seen else len(cpus)
    what = "%d CPUs in the affinity mask, %s" % (len(cpus), "%d distinct (package, core) pairs" % n if seen else how)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                fields = f.read().split()
            if path.endswith("cpu.max"):
                quota = None if fields[0] == "max" else float(fields[0]) / float(fields[1])
            else:
                q = float(fields[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                    quota = None if q <= 0 else q / float(g.read().split()[0])
            if quota is not None and quota < n:
                n = max(1, int(quota))
                what += ", cgroup quota %.1f CPUs" % quota
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n), what


def cpu_baseline(tables, col, nz, nzen, albedo, olr_gpu, rad=None, photon_scale=None):
    """The oracle (port of the reference algorithm, OpenMP over bins like the reference's
    `!$omp parallel do`, /root/reference/src/radtran/clima_radtran_types.f90:638-640 and
    clima_radtran_radiate.f90:50-52) on this box's host cores: whole radiate() calls of the same workload,
    bounded to ~25 s in all.  SURVEY 8(d): timed at ONE thread (the reference's Python default,
    clima/__init__.py:2) and at ALL PHYSICAL cores (`value`, `cores`); 16 threads are timed as well (the figure
    of rounds 1-3, and the share of a 1-GPU box that its worker pools are sized for)."""
    from oracle import oracle as O
    O.build()
    cores, cores_how = physical_cores()
    o = O.OracleRadtran(tables, nz, nzen, albedo)
    if photon_scale is not None:
        o.set_scalars(photon_scale_factor=photon_scale)

    def timed(threads, budget, cap):
        O.lib().orc_set_num_threads(threads)
        o.radiate(*col.args())  # warm (page-in, thread pool)
        n, t0 = 0, time.perf_counter()
        while True:
            o.radiate(*col.args())
            n += 1
            el = time.perf_counter() - t0
            if el > budget or n >= cap:
                return n, el

    n, el = timed(cores, 8.0, 200)
    n16 = el16 = None
    if cores != 16 and cores > 16:
        n16, el16 = timed(16, 5.0, 40)
    n1, el1 = timed(1, 6.0, 6)
    O.lib().orc_set_num_threads(cores)
    _, olr_o = o.TOA_fluxes(*col.args())
    # the Jacobian's unit of work on the CPU: one IR-only call on the stored opacities (clima_adiabat_solve.f90:811-812)
    t0 = time.perf_counter()
    for _ in range(4):
        o.radiate(*col.args(), compute_solar=False, compute_opacity=False)
    ir_only_ms = 1e3 * (time.perf_counter() - t0) / 4
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    out = {"cpu_baseline": {"value": n / el, "unit": "calls/s", "cores": cores, "kind": "port",
                            "sample": "%d whole radiate() calls of the same workload (%.1f s)" % (n, el),
                            "cores_how": cores_how,
                            "single_thread": {"value": n1 / el1, "unit": "calls/s",
                                              "sample": "%d calls (%.1f s)" % (n1, el1)},
                            "threads_16": ({"value": n16 / el16, "unit": "calls/s", "sample": "%d calls (%.1f s)" % (n16, el16)}
                                           if n16 else None),
                            "ir_only_call_ms": ir_only_ms,
                            "cpu_model": model, "host_cpus": os.cpu_count()},
           "olr_rel_err_vs_cpu": abs(olr_gpu - olr_o) / abs(olr_o)}
    if rad is not None:  # level fluxes of both channels and the planetary albedo
        import numpy as np
        errs = []
        for wg, wo in ((rad.wrk_ir, o.wrk_ir), (rad.wrk_sol, o.wrk_sol)):
            for a, b in ((wg.fup_n, wo.fup_n), (wg.fdn_n, wo.fdn_n)):
                a, b = np.asarray(a), np.asarray(b)
                errs.append(float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
        alb_g = rad.wrk_sol.fup_n[nz] / rad.wrk_sol.fdn_n[nz]
        alb_o = o.wrk_sol.fup_n[nz] / o.wrk_sol.fdn_n[nz]
        out["max_level_flux_err_vs_cpu"] = max(errs)
        out["albedo"] = float(alb_g)
        out["albedo_rel_err_vs_cpu"] = float(abs(alb_g - alb_o) / abs(alb_o))
    return out


if __name__ == "__main__":
    main()
