#!/bin/bash
# Sanitizer pass over everything that runs on the HOST (run here, in the build container: no GPU needed, and GPU
# sanitizers are not available on this pool).  The reference's CI runs its test driver under valgrind
# (.github/workflows/test.yaml:48-55); this is the counterpart:
#   1. oracle/clima_oracle.c with gcc -fsanitize=address,undefined (+ leak check): a C driver through the whole
#      orc_* API, then the CPU test-suite's oracle tests against that build;
#   2. the host side of libclima_radtran_hip.so with hipcc's host-only AddressSanitizer + UBSan
#      (-fsanitize=address,undefined -fno-gpu-sanitize): a C driver through construction, every validation / error
#      path, getters / setters and destruction, then tests/test_abi.py against that build.
# Writes profiles/r04_sanitize.log; exit code 0 = no report.
set -o pipefail
cd "$(dirname "$0")/.."
LOG=profiles/r04_sanitize.log
: > $LOG
say() { echo "== $*" | tee -a $LOG; }
fail() { echo "SANITIZE: FAILED at: $*" | tee -a $LOG; exit 1; }
export ASAN_OPTIONS=detect_leaks=1:halt_on_error=1:abort_on_error=0
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
SANFLAGS="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined"

say "1a. oracle: gcc $SANFLAGS, C driver through the orc_* API (leak check on)"
mkdir -p oracle/_san
gcc $SANFLAGS -fopenmp -ffp-contract=off -std=c11 -Wall -Wextra tools/san/oracle_driver.c oracle/clima_oracle.c -o oracle/_san/oracle_driver -lm 2>&1 | tee -a $LOG || fail "oracle driver build"
OMP_NUM_THREADS=4 ./oracle/_san/oracle_driver 2>&1 | tee -a $LOG || fail "oracle driver"

say "1b. oracle: the CPU suite's oracle tests against the sanitizer build (python: interpreter leaks are not ours, leak check off)"
gcc $SANFLAGS -fPIC -fopenmp -ffp-contract=off -std=c11 -shared -o oracle/_san/liborc.so oracle/clima_oracle.c -lm 2>&1 | tee -a $LOG || fail "liborc asan build"
gcc $SANFLAGS -fPIC -fopenmp -mfma -mavx2 -ffp-contract=fast -std=c11 -shared -o oracle/_san/liborc_fma.so oracle/clima_oracle.c -lm 2>&1 | tee -a $LOG || fail "liborc_fma asan build"
CLIMA_ORACLE_DIR=$PWD/oracle/_san ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  python3 -m pytest tests/test_oracle_units.py tests/test_oracle_golden.py tests/test_atmosphere.py tests/test_data_loader.py -x -q -p no:cacheprovider 2>&1 | tail -4 | tee -a $LOG
[ ${PIPESTATUS[0]} -eq 0 ] || fail "oracle tests under ASan"

say "2a. library host side: hipcc host-only ASan + UBSan build"
mkdir -p clima_amd/csrc/_san
RT=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
/opt/rocm/bin/hipcc $SANFLAGS -fno-gpu-sanitize -shared-libasan --offload-arch=gfx950 -std=c++17 -fPIC -shared -w \
  -mllvm -instcombine-max-copied-from-constant-users=100000 clima_amd/csrc/kernels.hip clima_amd/csrc/radtran_api.hip clima_amd/csrc/radtran_loader.hip \
  -o clima_amd/csrc/_san/libclima_radtran_hip.so -L/opt/rocm/lib -lrccl -ldl -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$RT 2>&1 | tee -a $LOG || fail "library asan build"

say "2b. library host side: C driver (construction, validation and error paths, getters, destruction; leak check on)"
cat > clima_amd/csrc/_san/lsan.supp <<SUPP
# one-time allocations of the HIP / HSA / RCCL runtimes at load (not ours, never freed by design)
leak:libamdhip64
leak:libhsa-runtime64
leak:librccl
leak:librocprofiler
SUPP
/opt/rocm/lib/llvm/bin/clang $SANFLAGS -shared-libasan -std=c11 -Wall tools/san/abi_driver.c -o clima_amd/csrc/_san/abi_driver \
  -Lclima_amd/csrc/_san -lclima_radtran_hip -Wl,-rpath,$PWD/clima_amd/csrc/_san -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$RT -lm 2>&1 | tee -a $LOG || fail "abi driver build"
LSAN_OPTIONS=suppressions=$PWD/clima_amd/csrc/_san/lsan.supp:print_suppressions=0 ./clima_amd/csrc/_san/abi_driver 2>&1 | tee -a $LOG
[ ${PIPESTATUS[0]} -eq 0 ] || fail "abi driver"

say "2c. library host side: tests/test_abi.py against the sanitizer build"
CLIMA_HIP_LIB=$PWD/clima_amd/csrc/_san/libclima_radtran_hip.so ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  LD_PRELOAD="$RT/libclang_rt.asan-x86_64.so" \
  python3 -m pytest tests/test_abi.py -x -q -p no:cacheprovider 2>&1 | tail -4 | tee -a $LOG
[ ${PIPESTATUS[0]} -eq 0 ] || fail "test_abi under ASan"
say "SANITIZE: clean"
