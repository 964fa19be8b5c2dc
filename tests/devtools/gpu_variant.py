#!/usr/bin/env python3
"""Developer tool: time config 2 and check opr/flux parity for an alternative build of the
library (experiments compiled with extra -D flags into csrc/<name>.so).
Usage: gpu_variant.py <lib file name in clima_amd/csrc> [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from clima_amd import lib
name = sys.argv[1]
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), name)
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
from oracle import oracle as O
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tb = S.modern_earth_tables()
col = S.modern_earth_column(200)
r = Radtran(tb, 200, 8, 0.15)
o = O.OracleRadtran(tb, 200, 8, 0.15)
isr, olr = r.TOA_fluxes(*col.args())
isr_o, olr_o = o.TOA_fluxes(*col.args())
rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(np.abs(np.asarray(b)), 1e-300)))
opr = [rel(a, b) for a, b in zip(r.opr(), o.opr())]
r.upload_column(*col.args())
best = 1e9
for rep in range(3):
    for _ in range(10): r.radiate_resident()
    r.synchronize()
    t0 = time.time()
    for _ in range(reps): r.radiate_resident()
    r.synchronize()
    best = min(best, (time.time() - t0) / reps)
r.profile(True); r.profile_reset()
for _ in range(50): r.radiate_resident()
r.synchronize()
ks = [r.kernel_time(i) for i in range(4)]
print("%s: %.1f us/call (best of 3x%d) | prep %.2f integrate %.2f | opacity %.1f twostream %.1f us | OLR rel %.1e ISR rel %.1e opr rel %s" % (
    name, best * 1e6, reps, 1e3 * ks[0][0] / max(ks[0][1], 1), 1e3 * ks[3][0] / max(ks[3][1], 1), 1e3 * ks[1][0] / max(ks[1][1], 1), 1e3 * ks[2][0] / max(ks[2][1], 1),
    abs(olr - olr_o) / abs(olr_o), abs(isr - isr_o) / abs(isr_o), " ".join("%.1e" % x for x in opr)))
