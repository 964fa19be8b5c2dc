#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes (gpurun_out/<prefix>_<set>/**/counter_collection.csv) into a
markdown table under profiles/.  Usage: pmc_summary.py <prefix> <out.md> [title]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    prefix, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else "PMC summary"
    lines = ["# %s" % title, "",
             "One `rocprofv3 --kernel-trace --pmc <set>` pass per section (counters never combined with",
             "tracing domains other than the kernel trace).  Values are per-launch means over all launches",
             "of each kernel.  FETCH_SIZE / WRITE_SIZE are KiB as reported; per MI355X_MICROARCH.md",
             "FETCH_SIZE under-counts wide coalesced reads by up to 2x on gfx950, so read-side HBM bytes",
             "lie between 1x and 2x the reported figure.", ""]
    for d in sorted(glob.glob(os.path.join("gpurun_out", prefix + "_*"))):
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        per_dispatch = defaultdict(lambda: defaultdict(float))
        names = {}
        for fn in files:
            with open(fn) as f:
                for row in csv.DictReader(f):
                    k = row["Kernel_Name"]
                    if not k.startswith("clima::") and "clima::" not in k:
                        continue
                    k = k.replace("void ", "").split("(")[0]
                    key = (fn, row["Dispatch_Id"])
                    names[key] = k
                    per_dispatch[key][row["Counter_Name"]] += float(row["Counter_Value"])
        for key, ctr in per_dispatch.items():
            for c, v in ctr.items():
                acc[names[key]][c].append(v)
        lines += ["## %s" % os.path.basename(d), "", "| kernel | launches | counters (mean per launch) |", "|---|---|---|"]
        for k, ctr in acc.items():
            n = max(len(v) for v in ctr.values())
            lines.append("| `%s` | %d | %s |" % (k, n, ", ".join(
                "%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(ctr.items()))))
        lines.append("")
    with open(out, "w") as f:
        f.write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
