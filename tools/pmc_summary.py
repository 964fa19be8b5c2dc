#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes (gpurun_out/<prefix>_<set>/**/counter_collection.csv) into a
markdown table under profiles/.  Usage: pmc_summary.py <prefix> <out.md> [title] [--json out.json workload_key]

With --json the per-kernel counters also go into a small JSON record next to the SHA-256 of
clima_amd/csrc/kernels.hip at the time of the pass: bench.py prints PMC-derived figures (traffic,
FP64 issue fraction) only while the kernels still have that hash."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    argv = list(sys.argv)
    js = None
    if "--json" in argv:
        i = argv.index("--json")
        js = (argv[i + 1], argv[i + 2])
        del argv[i:i + 3]
    prefix, out = argv[1], argv[2]
    title = argv[3] if len(argv) > 3 else "PMC summary"
    record = {}
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
            short = "k_" + k.split("clima::k_")[-1].split("<")[0] if "clima::k_" in k else k
            rec = record.setdefault(short, {})
            for c, v in ctr.items():
                rec[{"FETCH_SIZE": "FETCH_SIZE_KiB", "WRITE_SIZE": "WRITE_SIZE_KiB"}.get(c, c)] = sum(v) / len(v)
        lines.append("")
    with open(out, "w") as f:
        f.write("\n".join(lines))
    print("\n".join(lines))
    if js:
        import hashlib
        import json
        path, key = js
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
        with open(os.path.join(root, "clima_amd", "csrc", "kernels.hip"), "rb") as f:
            h = hashlib.sha256(f.read()).hexdigest()
        try:
            with open(path) as f:
                data = json.load(f)
        except OSError:
            data = {}
        if data.get("kernels_hip_sha256") != h:
            data = {"kernels_hip_sha256": h, "workloads": {}}
        data["source"] = os.path.relpath(out, root) if os.path.isabs(out) else out
        data["workloads"][key] = record
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
        print("wrote", path)


if __name__ == "__main__":
    main()
