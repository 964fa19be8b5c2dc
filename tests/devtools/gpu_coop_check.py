import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
from oracle import oracle as O
tb = S.modern_earth_tables(nw=int(sys.argv[1]) if len(sys.argv) > 1 else 100)
nz = 200
col = S.modern_earth_column(nz)
o = O.OracleRadtran(tb, nz, 8, 0.15); o.radiate(*col.args())
res = {}
for name, items in (("lane", "0"), ("coop", "100000000")):
    os.environ["CLIMA_HIP_COOP_ITEMS"] = items
    r = Radtran(tb, nz, 8, 0.15)
    r.radiate(*col.args())
    res[name] = (r.opr(), np.array(r.f_total), r.wrk_ir.fup_n, r.wrk_sol.fdn_n)
    for a, b, nm in zip(r.opr(), o.opr(), ("tau", "w0", "g", "tau_band")):
        print(name, nm, "max rel err vs oracle %.2e" % np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
    print(name, "f_total scaled err %.2e" % (np.max(np.abs(np.array(r.f_total) - o.f_total)) / np.max(np.abs(o.f_total))))
for a, b, nm in zip(res["lane"][0], res["coop"][0], ("tau", "w0", "g", "tau_band")):
    print("lane vs coop", nm, "%.2e" % np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
print("lane vs coop f_total %.2e" % (np.max(np.abs(res["lane"][1] - res["coop"][1])) / np.max(np.abs(res["lane"][1]))))
