#!/usr/bin/env python3
"""Benchmark of the radiate() hot path (BASELINE.json: `radiate() calls/sec`, 200-layer
ModernEarth column, full solar+IR correlated-k bin grid).

  python bench.py --gpus N --steps K --warmup W

A "step" is ONE full `Radtran%radiate` call -- opacity assembly + IR + solar two-stream +
spectral integration -- on the ModernEarth column of tests/test_radtran.f90 (config 2:
nz=200, nw=1000 opacity bins (600 IR / 600 solar), 8 g-points, 8 zenith angles, 5
k-distribution species, synthetic tables: SURVEY.md 8(d)).  Inputs (tables and the
column) are resident in HBM before the timed region starts; the PCIe-inclusive rate is
reported separately in DESIGN.md.

N > 1 (launched by torch.distributed.run, one rank per GPU): the spectral bins of the SAME
call are sharded over the ranks (work-balanced contiguous ranges) and every step ends with
one RCCL all-reduce of the 4*(nz+1) partial level fluxes -- strong scaling of one call.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed with HIP
events on the stream it is launched on; `cpu_baseline` is the oracle (a port of the
reference's algorithm, OpenMP over bins exactly like the reference) timed on this box's
host cores on the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
KERNELS = ["prep", "opacity", "twostream", "integrate"]
EVENT_STRIDE = 16  # HIP events around the dominant kernel on every 16th launch of the timed region (first one included)
# HBM bytes of one k_opacity8 launch on this exact workload from the PMC passes committed under
# profiles/ (FETCH_SIZE + WRITE_SIZE, KiB -> bytes; bench.py cannot collect counters itself)
PMC_TRAFFIC_BYTES = {"fused": (2.732e4 + 4.737e4) * 1024.0,      # profiles/r01j_pmc_summary.md, k_fused
                     "opacity": (1.640e4 + 4.681e4) * 1024.0}    # profiles/r01d_pmc_summary.md, k_opacity8
PMC_TRAFFIC_SOURCE = "FETCH_SIZE + WRITE_SIZE (KiB) of the dominant kernel per launch: profiles/r01j_pmc_summary.md (k_fused), profiles/r01d_pmc_summary.md (k_opacity8)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--nz", type=int, default=200)
    ap.add_argument("--nzen", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
        raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
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

    nz, nzen = args.nz, args.nzen
    tables = S.modern_earth_tables()
    col = S.modern_earth_column(nz)
    rad = Radtran(tables, nz, nzen, 0.15)  # tests/test_radtran.f90:35-38
    if dist_on:
        rad.set_bin_shard(rank, world)
        fake = os.environ.get("CLIMA_BENCH_FAKE_SHARD")   # "rank,world": rehearse one rank's share of an N-GPU step on one GPU
        if fake and world == 1:
            rad.set_bin_shard(*[int(x) for x in fake.split(",")])
    rad.upload_column(*col.args())
    flux = rad.flux_tensor() if dist_on else None
    # The all-reduce is ordered against the library's kernels on the device: the library's HIP
    # stream is made torch's current stream for the collective, so RCCL's stream waits for the
    # partial fluxes and the library stream waits for the reduced ones -- no host round trip
    # inside a step.  CLIMA_BENCH_HOST_SYNC=1 selects the plain host-synchronised form.
    host_sync = os.environ.get("CLIMA_BENCH_HOST_SYNC") == "1"
    lib_stream = None
    if dist_on and not host_sync:
        try:
            lib_stream = torch.cuda.ExternalStream(rad.stream())
        except Exception as e:  # same torch build on every rank: all of them take the same branch
            print("bench: torch.cuda.ExternalStream unavailable (%s); host-synchronised steps" % e, file=sys.stderr)
            host_sync = True

    def step():
        rad.radiate_resident()
        if dist_on:
            if host_sync:
                rad.synchronize()                 # library stream -> host
                dist.all_reduce(flux)             # RCCL over xGMI: 4*(nz+1) doubles
                torch.cuda.current_stream().synchronize()
            else:
                with torch.cuda.stream(lib_stream):
                    dist.all_reduce(flux)
            rad.finish_reduced()              # f_total from the reduced fluxes

    def barrier():
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
    rad.profile(2)
    rad.profile_stride(EVENT_STRIDE)
    rad.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    kt_dom = rad.kernel_time(1)
    if kt_dom[1] > 0:
        kt[1] = kt_dom
    rad.profile_stride(1)
    rad.profile(False)

    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    # ---- companion figure for N > 1 (not `value`): the column-parallel form of config 4 --
    # every rank runs whole, unsharded calls on its own column, no collective (weak scaling)
    col_par = None
    if dist_on:
        rad2 = Radtran(tables, nz, nzen, 0.15)
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
    isr = float((flux[3 * (nz + 1) + nz] - flux[2 * (nz + 1) + nz]).item()) if dist_on else None
    if dist_on:
        olr = -float((flux[1 * (nz + 1) + nz] - flux[0 * (nz + 1) + nz]).item())
    else:
        w_ir, w_sol = rad.wrk_ir, rad.wrk_sol
        olr = -(w_ir.fdn_n[nz] - w_ir.fup_n[nz])
        isr = w_sol.fdn_n[nz] - w_sol.fup_n[nz]

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = args.steps / dt
        names = list(KERNELS)
        if rad.fused and kt[2][1] == 0:   # opacity and two-stream work ran as one grid (k_fused), timed as kernel id 1
            names[1] = "fused"
        per_kernel_us = {k: (1e3 * ms / n if n else 0.0) for k, (ms, n) in zip(names, kt) if n}
        dom = max(per_kernel_us, key=per_kernel_us.get)
        ab = rad.algorithmic_bytes()
        # algorithmic bytes of one call: distinct table nodes + inputs + outputs,
        # prorated to the bins this rank owns
        frac = rad.bin_shard()[1] / float(tables.nw)
        b_alg = (ab["tables_distinct"] + ab["output"]) * frac + ab["input"]
        dur = per_kernel_us[dom] * 1e-6
        roofline = {"bound": "hbm", "kernel": "k_" + dom, "achieved": b_alg / dur / 1e9 if dur > 0 else 0.0,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (b_alg / dur / 1e9) / HBM_PEAK_GBS if dur > 0 else 0.0,
                    "traffic": PMC_TRAFFIC_BYTES.get(dom) if (not dist_on and (nz, nzen) == (200, 8)) else None,
                    "traffic_source": PMC_TRAFFIC_SOURCE,
                    "algorithmic_bytes": b_alg, "kernel_us": per_kernel_us,
                    "whole_call_frac": (b_alg / (dt / args.steps) / 1e9) / HBM_PEAK_GBS}
        out = {"metric": "radiate() calls/sec", "value": value, "unit": "calls/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": "ModernEarth column, one Radtran%%radiate call: nz=%d, nw=1000 "
                                      "(600 IR + 600 solar bins), 8 g-points, %d zenith angles, nk=5; "
                                      "config 2 of BASELINE.json" % (nz, nzen),
                          "parallelism": ("bins sharded over %d GPUs + 1 all-reduce of %d f64" % (world, 4 * (nz + 1)))
                          if world > 1 else "1 GPU"},
               "olr_W_m2": olr / 1e3, "isr_W_m2": isr / 1e3, "roofline": roofline}
        if col_par is not None:
            out["column_parallel"] = col_par
        if not args.no_cpu_baseline and world == 1:
            out.update(cpu_baseline(tables, col, nz, nzen, olr, rad))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist_on:
        dist.destroy_process_group()


def cpu_baseline(tables, col, nz, nzen, olr_gpu, rad=None):
    """The oracle (port of the reference algorithm, OpenMP over bins like the reference's
    `!$omp parallel do`) on this box's host cores: whole radiate() calls of the same
    workload, bounded to ~10-30 s in all.  Timed at all usable cores (the figure in `value`)
    and at one thread (the reference's Python default, clima/__init__.py:2)."""
    from oracle import oracle as O
    O.build()
    cores = min(os.cpu_count() or 1, 16)
    o = O.OracleRadtran(tables, nz, nzen, 0.15)

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

    n, el = timed(cores, 10.0, 40)
    n1, el1 = timed(1, 8.0, 8)
    O.lib().orc_set_num_threads(cores)
    _, olr_o = o.TOA_fluxes(*col.args())
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    out = {"cpu_baseline": {"value": n / el, "unit": "calls/s", "cores": cores, "kind": "port",
                            "sample": "%d whole radiate() calls of the same workload (%.1f s)" % (n, el),
                            "single_thread": {"value": n1 / el1, "unit": "calls/s",
                                              "sample": "%d calls (%.1f s)" % (n1, el1)},
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
