import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import test_gpu_fuzz as F
from oracle import oracle as O
from clima_amd import synthetic as S
from clima_amd.lib import load
L = load()
seed = int(sys.argv[1])
tb, nz, nzen, albedo, col, scalars, rng = F._case(5000 + seed)
print("nz", nz, "scalars", scalars, "ng", tb.get("ng") if isinstance(tb, dict) else None)
r, o = F._pair(O, tb, nz, nzen, albedo, **scalars)
r.radiate(*col.args()); o.radiate(*col.args())
ncol = int(rng.integers(2, 40))
T = np.repeat(np.asarray(col["T"], dtype=float)[:, None], ncol, axis=1)
Ts = np.full(ncol, float(col["T_surface"]))
for c in range(ncol):
    for _ in range(int(rng.integers(0, 12))):
        j = int(rng.integers(0, nz + 1))
        d = float(rng.choice([1e-6, 1e-3, 0.1, 3.0, 40.0])) * float(rng.choice([-1.0, 1.0]))
        if j == nz: Ts[c] += d
        else: T[j, c] = max(T[j, c] + d, 5.0)
r.ir_green = 0
gen = r.radiate_ir_batch(Ts, T)
res = {}
for form in (1, 0):
    L.clima_test_green_far_form_set(C.byref(C.c_int(form)))
    r.ir_green = 2
    res[form] = r.radiate_ir_batch(Ts, T)
print("ncol", ncol, "batches", r.ir_green_batches)
for c in range(ncol):
    w = S.Column(col); w["T"] = T[:, c].copy(); w["T_surface"] = Ts[c]
    o.radiate(*w.args(), compute_solar=False, compute_opacity=False)
    want = [np.array(o.wrk_ir.fup_n), np.array(o.wrk_ir.fdn_n), np.array(o.f_total)]
    ndev = int(np.sum(T[:, c] != T[:, 0])) 
    line = "col %2d:" % c
    for i in range(3):
        sc = max(float(np.max(np.abs(want[i]))), 1e-300)
        line += "  [%d] vec %.1e mfma %.1e gen %.1e" % (i, np.max(np.abs(res[1][i][:, c] - want[i])) / sc, np.max(np.abs(res[0][i][:, c] - want[i])) / sc, np.max(np.abs(gen[i][:, c] - want[i])) / sc)
    print(line)
w = S.Column(col); w["T"] = T[:, 0].copy(); w["T_surface"] = Ts[0]
o2 = O.OracleRadtran(tb, nz, nzen, albedo, variant="fma"); o2.set_scalars(**scalars); o2.radiate(*col.args())
o.radiate(*w.args(), compute_solar=False, compute_opacity=False); o2.radiate(*w.args(), compute_solar=False, compute_opacity=False)
print("max |fup_n| %.3e  max |fdn_n| %.3e  max |f_total| %.3e" % (np.max(np.abs(o.wrk_ir.fup_n)), np.max(np.abs(o.wrk_ir.fdn_n)), np.max(np.abs(o.f_total))))
print("oracle plain vs fma: fup %.2e fdn %.2e (absolute)" % (np.max(np.abs(np.array(o.wrk_ir.fup_n) - np.array(o2.wrk_ir.fup_n))), np.max(np.abs(np.array(o.wrk_ir.fdn_n) - np.array(o2.wrk_ir.fdn_n)))))
print("hip vs oracle: fup %.2e fdn %.2e (absolute)" % (np.max(np.abs(res[0][0][:, 0] - np.array(o.wrk_ir.fup_n))), np.max(np.abs(res[0][1][:, 0] - np.array(o.wrk_ir.fdn_n)))))
r.radiate(*w.args(), compute_solar=False, compute_opacity=False)
print("hip single call vs oracle: fup %.2e fdn %.2e (absolute)" % (np.max(np.abs(np.array(r.wrk_ir.fup_n) - np.array(o.wrk_ir.fup_n))), np.max(np.abs(np.array(r.wrk_ir.fdn_n) - np.array(o.wrk_ir.fdn_n)))))
