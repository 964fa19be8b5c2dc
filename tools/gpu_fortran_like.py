#!/usr/bin/env python3
"""Developer diagnostic: what the Fortran module's habit of pushing every public field before every radiate costs
(clima_amd/fortran/clima_radtran_hip.f90 push_fields, emulated through the same C setters).  Round 3: the setters
mark the device copies stale only when a value changes -- before, every call paid a stream synchronise, six
allocations and six copies (298 against 130 us per synchronous call)."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
tb = S.modern_earth_tables()
col = S.modern_earth_column(200)
r = Radtran(tb, 200, 8, 0.15)
a = col.args()
zu, zw, alb, em = np.array(r.zenith_u), np.array(r.zenith_weights), np.array(r.surface_albedo), np.array(r.surface_emissivity)
def push():   # what clima_amd/fortran/clima_radtran_hip.f90 push_fields does before every radiate
    r.zenith_u = zu; r.zenith_weights = zw; r.surface_albedo = alb; r.surface_emissivity = em
    r.has_hard_surface = True; r.photon_scale_factor = 1.0; r.ir_tau_min = 1e-6; r.diurnal_fac = 0.5
for _ in range(5):
    push(); r.TOA_fluxes(*a)
ts = []
for _ in range(50):
    t0 = time.perf_counter(); push(); r.TOA_fluxes(*a); ts.append(time.perf_counter() - t0)
print("Fortran-shim-like call (public fields pushed before every radiate): median %.1f us" % (1e6 * np.median(ts)))
ts = []
for _ in range(50):
    t0 = time.perf_counter(); r.TOA_fluxes(*a); ts.append(time.perf_counter() - t0)
print("without the push: median %.1f us" % (1e6 * np.median(ts)))
