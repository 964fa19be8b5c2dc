#!/usr/bin/env python3
"""Developer diagnostic: scratch (register-spill) instructions per kernel of kernels.hip, from the gfx950
assembly.  A wave of these kernels stalls for an L2 round trip on every scratch reload, so builds that differ
only in where the register allocator spilled differ by several us per call: the hot
kernels should show no scratch instruction at all between their first and last lines.

  python tools/scratch_report.py [-D...] [regex on the mangled name]     (cross-compiles: no GPU needed, ~1 min)
"""
import os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "clima_amd", "csrc", "kernels.hip")
flags = [a for a in sys.argv[1:] if a.startswith("-")]
pat = ([a for a in sys.argv[1:] if not a.startswith("-")] + ["k_"])[0]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only",
                           "-w", "-mllvm", "-instcombine-max-copied-from-constant-users=100000", src, "-o", out] + flags)
    name, n, rows = None, 0, []
    for line in open(out):
        m = re.match(r"^(_ZN5clima\w+):", line)
        if m:
            name, n, ld, st, where = m.group(1), 0, 0, 0, []
            continue
        if name is None:
            continue
        n += 1
        if "scratch_load" in line:
            ld += 1; where.append(n)
        elif "scratch_store" in line:
            st += 1; where.append(n)
        elif line.startswith(".Lfunc_end"):
            if re.search(pat, name):
                rows.append((name, n, ld, st, where))
            name = None
if not rows:
    sys.exit("no kernel matches %r" % pat)
dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
for (name, n, ld, st, where), dn in zip(rows, dem):
    short = dn.split("(")[0].replace("void clima::", "")
    print("%-46s %6d lines  %3d scratch loads %3d stores  at %s" % (short, n, ld, st, " ".join(map(str, where[:24])) + (" ..." if len(where) > 24 else "")))
