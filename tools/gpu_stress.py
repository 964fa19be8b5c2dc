#!/usr/bin/env python3
"""Developer tool: repeat config 2 many times and require bitwise-identical level fluxes and
spectra every time (a stale read in the fused kernel's block-to-block hand-off would show up
as a difference).  Usage: gpu_stress.py [calls]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
tb = S.modern_earth_tables()
col = S.modern_earth_column(200)
r = Radtran(tb, 200, 8, 0.15)
r.upload_column(*col.args())
def snap():
    r.radiate_resident(); r.synchronize()
    return [np.array(x).copy() for x in (r.wrk_ir.fup_n, r.wrk_ir.fdn_n, r.wrk_sol.fup_n, r.wrk_sol.fdn_n, r.f_total)]
ref = snap()
spec_ref = [np.array(x).copy() for x in (r.wrk_ir.fup_a, r.wrk_sol.fdn_a, r.wrk_sol.amean)]
bad = 0
for i in range(n):
    cur = snap()
    if any(not np.array_equal(a, b) for a, b in zip(cur, ref)):
        bad += 1
    if i % 500 == 0:
        spec = [np.array(x) for x in (r.wrk_ir.fup_a, r.wrk_sol.fdn_a, r.wrk_sol.amean)]
        if any(not np.array_equal(a, b) for a, b in zip(spec, spec_ref)):
            bad += 1
        print("call", i, "mismatches so far", bad, flush=True)
print("calls %d mismatching %d" % (n, bad))
sys.exit(1 if bad else 0)
