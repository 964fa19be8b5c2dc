#!/usr/bin/env python3
"""Developer diagnostic: GPU-vs-oracle flux error of fuzz cases as a function of ir_tau_min."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("fz", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "test_gpu_fuzz.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from clima_amd.radtran import Radtran
from oracle import oracle as O
for seed in [int(a) for a in sys.argv[1:]] or [6, 7, 16, 26]:
    tb, nz, nzen, albedo, col, scalars, rng = m._case(seed)
    for tm in (scalars["ir_tau_min"], 1e-6):
        sc = dict(scalars, ir_tau_min=tm)
        r = Radtran(tb, nz, nzen, albedo); o = O.OracleRadtran(tb, nz, nzen, albedo)
        for k, v in sc.items(): setattr(r, k, v)
        o.set_scalars(**sc)
        isr, olr = r.TOA_fluxes(*col.args()); isr_o, olr_o = o.TOA_fluxes(*col.args())
        e = max(float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b))) for a, b in
                ((r.wrk_ir.fup_n, o.wrk_ir.fup_n), (r.wrk_ir.fdn_n, o.wrk_ir.fdn_n)))
        tau = r.opr()[0]
        print("seed %d ir_tau_min %.1e: OLR rel %.1e  IR level err %.1e  min tau %.1e" % (seed, tm, abs(olr - olr_o) / abs(olr_o), e, float(tau.min())))
