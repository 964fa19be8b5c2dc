#!/usr/bin/env python3
"""Developer diagnostic: where a SMALL call's time goes (AdiabatClimate's template shape: 102-layer doubled grid, 4 zenith
angles, `nw` bins).  Run under `rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 tools/gpu_small_call_trace.py 200`;
then `python3 tools/gpu_small_call_trace.py --read DIR/run_kernel_trace.csv` prints kernel durations and the gaps between them
in the steady-state part (resident calls enqueued back to back)."""
import csv
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    import numpy as np
    rows = []
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void clima::", "").replace("clima::", "")))
    rows.sort()
    rows = rows[len(rows) // 2:]            # the second half: the 300 back-to-back resident calls
    names = [r[2] for r in rows]
    per = {}
    for (s, e, n), nxt in zip(rows, rows[1:] + [None]):
        per.setdefault(n, []).append((e - s, (nxt[0] - e) if nxt else 0))
    first = names.index(next(n for n in names if n.startswith("k_prep")))
    cyc = [n for n in names[first:first + 8]]
    print("kernel sequence:", cyc[:cyc[1:].index(cyc[0]) + 1] if cyc[0] in cyc[1:] else cyc)
    tot = 0.0
    for n, v in per.items():
        d = np.array([x[0] for x in v]) / 1e3
        g = np.array([x[1] for x in v]) / 1e3
        print("  %-60s n %4d  duration p50 %6.2f us   gap to the next kernel p50 %6.2f us" % (n[:60], len(v), np.median(d), np.median(g)))
        tot += np.median(d) + np.median(g)
    print("  sum of medians (duration + gap) %.1f us per call" % tot)
    sys.exit(0)
from clima_amd import synthetic as S
from clima_amd.atmosphere import copy_atm_to_radiative_grid
from clima_amd.radtran import Radtran
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tb = S.modern_earth_tables(nw=nw)
col = S.Column(copy_atm_to_radiative_grid(S.modern_earth_column(50)))
r = Radtran(tb, len(col["T"]), 4, 0.3)
r.upload_column(*col.args())
for _ in range(300):
    r.radiate_resident()
r.synchronize()
for _ in range(300):
    r.radiate_resident()
r.synchronize()
