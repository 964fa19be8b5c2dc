#!/usr/bin/env python3
"""Developer diagnostic: the synchronous drop-in call timed by a FORTRAN host (clima_amd/fortran/radtran_driver in its
`time` mode: rad%TOA_fluxes in a loop, system_clock around each call), config 2's size by default.
Usage: gpu_fortran_host.py [nz] [nw] [ncalls]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clima_amd import build, synthetic as S
from clima_amd.fortran_case import write_case
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n = sys.argv[3] if len(sys.argv) > 3 else "200"
exe = build.build_fortran_shim()
assert exe, "no Fortran compiler"
with tempfile.TemporaryDirectory() as d:
    case = os.path.join(d, "case.bin")
    write_case(case, S.modern_earth_tables(nw=nw), S.modern_earth_column(nz), 8, 0.15)
    out = subprocess.run([exe, case, os.path.join(d, "out.txt"), "time", n], capture_output=True, text=True, timeout=600)
    print("nz %d, nw %d, 8 zenith angles:" % (nz, nw))
    print(out.stdout.strip() or out.stderr.strip())
    if out.returncode:
        sys.exit(out.returncode)
    # the RCE Jacobian's batch from the same host (`jac` mode), on this grid and on AdiabatClimate's 402-layer doubled grid
    for nzj in (nz, 402):
        write_case(case, S.modern_earth_tables(nw=nw), S.modern_earth_column(nzj), 4, 0.15)
        out = subprocess.run([exe, case, os.path.join(d, "out.txt"), "jac"], capture_output=True, text=True, timeout=600)
        print(out.stdout.strip() or out.stderr.strip())
        if out.returncode:
            sys.exit(out.returncode)
