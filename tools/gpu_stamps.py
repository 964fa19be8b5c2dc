#!/usr/bin/env python3
"""Developer diagnostic: per-phase s_memtime stamps of one opacity wave (needs the
-DCLIMA_STAMPS build, clima_amd/csrc/libclima_radtran_hip_stamps.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from clima_amd import lib
lib.LIB_PATH = os.environ.get("CLIMA_STAMPS_LIB") or lib.LIB_PATH.replace("libclima_radtran_hip.so", "libclima_radtran_hip_stamps.so")
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg3 = os.environ.get("CLIMA_STAMPS_CONFIG") == "3"
tb = S.early_mars_tables() if cfg3 else S.modern_earth_tables()
col = S.early_mars_column(200) if cfg3 else S.modern_earth_column(200)
r = Radtran(tb, 200, 4 if cfg3 else 8, 0.2 if cfg3 else 0.15)
if world > 1: r.set_bin_shard(0, world)
r.upload_column(*col.args())
for _ in range(3): r.radiate_resident()
r.synchronize()
out = (C.c_longlong * (64 + 2 * 8192))()
r._L.clima_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
r._L.clima_debug_stamps(r._ptr, out)
s = list(out)
t0 = s[0]
print("continuum        %7d" % (s[1] - s[0]))
prev = s[1]
for sp in range(5):
    a = s[2 + 3 * sp]
    print("species %d interp %7d" % (sp, a - prev))
    if sp > 0:
        print("   sort          %7d" % (s[21 + sp] - a))
        print("   rebin         %7d" % (s[4 + 3 * sp] - s[21 + sp]))
        prev = s[4 + 3 * sp]
    else:
        prev = a
print("epilogue         %7d" % (s[20] - prev))
print("total            %7d  (s_memtime ticks = shader cycles)" % (s[20] - s[0]))

w = np.array(s[64:64 + 2 * 3128]).reshape(-1, 2).astype(float)
w = w[w[:, 1] > 0]
t0 = w[:, 0].min()
st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0   # s_memrealtime: 100 MHz -> us
print("waves %d: start us  p0 %.1f p50 %.1f p90 %.1f max %.1f" % (len(w), st.min(), np.median(st), np.percentile(st, 90), st.max()))
print("          end   us  p10 %.1f p50 %.1f p90 %.1f max %.1f" % (np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
print("          life  us  min %.1f p50 %.1f max %.1f" % ((en - st).min(), np.median(en - st), (en - st).max()))
hist, edges = np.histogram(st, bins=12)
print("start histogram:", list(zip(np.round(edges[:-1], 1), hist)))
late = st > 5
print("late starters: %d, their life p50 %.1f us; early life p50 %.1f us" % (late.sum(), np.median((en - st)[late]) if late.any() else 0, np.median((en - st)[~late])))

