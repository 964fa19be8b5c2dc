#!/usr/bin/env python3
"""Developer ablation timing of the nominal config (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clima_amd import synthetic as S
from clima_amd.radtran import Radtran
tb = S.modern_earth_tables()
col = S.modern_earth_column(200)
r = Radtran(tb, 200, 8, 0.15)
import sys as _s
if len(_s.argv) > 1: r.set_bin_shard(0, int(_s.argv[1]))
r.upload_column(*col.args())
names = ["prep", "opacity", "twostream", "integrate"]
for label, env in (("full", {}), ("ts: no thomas", {"CLIMA_HIP_DEBUG_SKIP_TS": "1"}), ("ts: 1 zenith", {"CLIMA_HIP_DEBUG_SKIP_TS": "2"}),
                   ("ts: neither", {"CLIMA_HIP_DEBUG_SKIP_TS": "3"}), ("op: no rorr", {"CLIMA_HIP_DEBUG_SKIP_OP": "1"})):
    for k in ("CLIMA_HIP_DEBUG_SKIP_TS", "CLIMA_HIP_DEBUG_SKIP_OP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    r.profile(True)
    for _ in range(3): r.radiate_resident()
    r.synchronize(); r.profile_reset()
    for _ in range(20): r.radiate_resident()
    r.synchronize()
    ks = [r.kernel_time(i) for i in range(4)]
    print("%-16s" % label, ", ".join("%s %.1f" % (n, 1e3 * ms / max(c, 1)) for n, (ms, c) in zip(names, ks)), flush=True)
