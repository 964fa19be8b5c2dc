import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
nz = int(sys.argv[1])
tb = S.modern_earth_tables(); col = S.modern_earth_column(nz)
r = Radtran(tb, nz, 8, 0.15); r.TOA_fluxes(*col.args())
n = 128
T = np.repeat(np.asarray(col["T"])[:, None], n, axis=1); Ts = np.full(n, float(col["T_surface"]))
for c in range(1, n): T[(c - 1) % nz, c] *= 1.01
r.radiate_ir_batch(Ts, T)
ts = []
for _ in range(12):
    t0 = time.time(); out = r.radiate_ir_batch(Ts, T); ts.append(time.time() - t0)
print("nz %d: %d columns, per call ms: %s -> best %.1f us/column" % (nz, n, " ".join("%.2f" % (t * 1e3) for t in ts), min(ts) / n * 1e6))

# config 4: many independent columns, one call each vs one batch
cols = S.perturbed_columns(256, nz, seed=7)
t0 = time.time()
for c in cols[:64]: r.TOA_fluxes(*c.args())
tl = (time.time() - t0) / 64
r.TOA_fluxes_batch(cols[:8])
t0 = time.time(); isr, olr = r.TOA_fluxes_batch(cols); tb = (time.time() - t0) / len(cols)
print("config 4 (nz %d): one TOA_fluxes per column %.1f us/column (%.0f columns/s); batch of %d: %.1f us/column (%.0f columns/s)"
      % (nz, tl * 1e6, 1 / tl, len(cols), tb * 1e6, 1 / tb))
