#!/usr/bin/env python3
"""Developer diagnostic: timeline of the fused grid (needs the -DCLIMA_STAMPS build,
clima_amd/csrc/libclima_radtran_hip_stamps.so): when opacity waves and two-stream blocks
start, become ready and end, and how many of each are resident over time."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from clima_amd import lib
lib.LIB_PATH = lib.LIB_PATH.replace("libclima_radtran_hip.so", "libclima_radtran_hip_stamps.so")
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
tb = S.modern_earth_tables()
col = S.modern_earth_column(200)
r = Radtran(tb, 200, 8, 0.15)
r.upload_column(*col.args())
for _ in range(3): r.radiate_resident()
r.synchronize()
out = (C.c_longlong * (64 + 2 * 8192))()
r._L.clima_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
r._L.clima_debug_stamps(r._ptr, out)
s = np.array(list(out), dtype=np.int64)
w = s[64:64 + 2 * 3128].reshape(-1, 2).astype(float)
w = w[w[:, 1] > 0]
t = s[64 + 2 * 3128:64 + 2 * 3128 + 3 * 2400].reshape(-1, 3).astype(float)
t = t[t[:, 2] > 0]
t0 = min(w[:, 0].min(), t[:, 0].min())
w = (w - t0) / 100.0; t = (t - t0) / 100.0     # us
print("opacity waves %d, two-stream blocks %d; grid ends at %.1f us" % (len(w), len(t), max(w[:, 1].max(), t[:, 2].max())))
print("two-stream block: wait p50 %.1f p90 %.1f max %.1f us; run p50 %.1f p90 %.1f us" % (
    np.median(t[:, 1] - t[:, 0]), np.percentile(t[:, 1] - t[:, 0], 90), (t[:, 1] - t[:, 0]).max(),
    np.median(t[:, 2] - t[:, 1]), np.percentile(t[:, 2] - t[:, 1], 90)))
print("  t(us)  opacity waves  ts blocks waiting  ts blocks running")
for x in np.arange(0, max(w[:, 1].max(), t[:, 2].max()) + 5, 5.0):
    ow = int(((w[:, 0] <= x) & (w[:, 1] > x)).sum())
    tw = int(((t[:, 0] <= x) & (t[:, 1] > x)).sum())
    tr = int(((t[:, 1] <= x) & (t[:, 2] > x)).sum())
    print("  %5.0f  %6d  %6d  %6d" % (x, ow, tw, tr))
names = ["", "setup (solar: tcum scan)", "coefficients + elimination", "upward sweep + affine map", "M7 suffix scan", "prefix scan + boundary", "level fluxes -> LDS", "barrier", "g-point sum + atomics"]
for label, base in (("stamped two-stream block A (whole-wave form: 1500; half-wave: a solar one)", 32), ("stamped two-stream block B (2200; an IR one)", 48)):
    v = s[base:base + 9].astype(float)
    if v[8] <= 0: continue
    print(label, "total %.0f cycles" % (v[8] - v[0]))
    for k in range(1, 9):
        print("   %-28s %7.0f" % (names[k] if k < len(names) else k, v[k] - v[k - 1]))
