"""Build the HIP C-ABI library in-tree: clima_amd/csrc/libclima_radtran_hip.so (gfx950)."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libclima_radtran_hip.so")
SOURCES = ["kernels.hip", "radtran_api.hip"]
DEPS = SOURCES + ["radtran_dev.h", "sort_network_64.inc", os.path.join("..", "..", "include", "clima_radtran_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, extra_flags=()):
    if not (force or is_stale()):
        return LIB
    cmd = [HIPCC] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra_flags=[a for a in sys.argv[1:] if a.startswith("-") and a != "--force"])
