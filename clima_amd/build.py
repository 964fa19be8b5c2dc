"""Build the HIP C-ABI library in-tree: clima_amd/csrc/libclima_radtran_hip.so (gfx950)."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libclima_radtran_hip.so")
SOURCES = ["kernels.hip", "radtran_api.hip", "radtran_loader.hip"]
DEPS = SOURCES + ["radtran_dev.h", "sort_network_64.inc", "rorr_xys_asm.inc", "ir_green.inc", os.path.join("..", "..", "include", "clima_radtran_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# The kernels take their parameter blocks by value; hipcc reads them straight from the kernel-argument
# segment (scalar loads) only while a block has at most `instcombine-max-copied-from-constant-users` uses
# (default 300) -- beyond that it copies the whole block to scratch at kernel entry and every access becomes
# a scratch load.  The unrolled opacity tile has more uses than that.
LLVM_FLAGS = ["-mllvm", "-instcombine-max-copied-from-constant-users=100000"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"] + LLVM_FLAGS
# RCCL: the bin-sharded step's all-reduce is issued by the library itself (radtran_comm_init_rank).  In a process
# that imported torch first, the loader resolves librccl.so.1 to the copy torch has already mapped (same SONAME).
LIBS = ["-L/opt/rocm/lib", "-lrccl", "-ldl", "-Wl,-rpath,/opt/rocm/lib"]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """`out`: write another build of the library there (A/B variants, tools/gpu_ab.sh) instead of LIB."""
    if not (force or is_stale() or out):
        return LIB
    cmd = [HIPCC] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out or LIB] + LIBS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out or LIB


FORTRAN_DIR = os.path.join(_HERE, "fortran")
FORTRAN_DRIVER = os.path.join(FORTRAN_DIR, "radtran_driver")
FORTRAN_FROM_FILES = os.path.join(FORTRAN_DIR, "radtran_from_files")   # the constructor-from-files host (radtran_from_files.f90)
FLANG = os.environ.get("FLANG", "/opt/rocm/bin/amdflang")


def build_fortran_shim(verbose=False):
    """Compile the ISO_C_BINDING shim module and the test_radtran-shaped driver against the
    HIP library (amdflang).  Returns the driver path, or None when no Fortran compiler."""
    if not os.path.exists(FLANG):
        if verbose:
            print("amdflang not found: skipping the Fortran shim")
        return None
    mod = os.path.join(FORTRAN_DIR, "clima_radtran_hip.f90")
    progs = [(FORTRAN_DRIVER, os.path.join(FORTRAN_DIR, "radtran_driver.f90")),
             (FORTRAN_FROM_FILES, os.path.join(FORTRAN_DIR, "radtran_from_files.f90"))]
    for exe, src in progs:
        if os.path.exists(exe) and all(os.path.getmtime(exe) > os.path.getmtime(x) for x in (mod, src, LIB)):
            continue
        cmd = [FLANG, "-O2", "-J", FORTRAN_DIR, mod, src, "-o", exe, "-L" + CSRC, "-lclima_radtran_hip",
               "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return FORTRAN_DRIVER


if __name__ == "__main__":
    # python -m clima_amd.build [--force] [-D...] [out=variants/lib_x.so]
    _out = next((a[4:] for a in sys.argv[1:] if a.startswith("out=")), None)
    build(force="--force" in sys.argv, verbose=True,
          extra_flags=[a for a in sys.argv[1:] if a.startswith("-") and a != "--force"], out=_out)
    if not _out:
        build_fortran_shim(verbose=True)
