"""AddressSanitizer + UndefinedBehaviorSanitizer + leak check on the CPU restatement (the oracle), through a C
driver that exercises the whole orc_* API (tools/san/oracle_driver.c).  The reference's CI runs its own driver
under valgrind (.github/workflows/test.yaml:48-55).  tools/sanitize.sh is the full pass (it also builds the HOST
side of the HIP library with hipcc's host-only ASan and runs tools/san/abi_driver.c and tests/test_abi.py against
it: minutes of compile time, so not part of this suite; its log is profiles/r03_sanitize.log)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_oracle_c_driver_under_asan_ubsan_lsan(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "oracle_driver")
    cmd = ["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fopenmp", "-ffp-contract=off",
           "-std=c11", os.path.join(ROOT, "tools", "san", "oracle_driver.c"), os.path.join(ROOT, "oracle", "clima_oracle.c"),
           "-o", exe, "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "asan" in (b.stderr or "").lower():
        pytest.skip("this gcc has no sanitizer runtime")
    assert b.returncode == 0, b.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "oracle_driver: ok" in r.stdout, r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr


def test_the_full_sanitizer_pass_left_a_clean_log():
    log = os.path.join(ROOT, "profiles", "r04_sanitize.log")
    assert os.path.exists(log), "run tools/sanitize.sh"
    text = open(log).read()
    assert "SANITIZE: clean" in text and "abi_driver: ok" in text and "oracle_driver: ok" in text
    assert "ERROR: AddressSanitizer" not in text and "runtime error:" not in text
    # nothing else may have shouted either (round 3's log held an RCCL "[FATAL ERROR]" that no check looked at)
    assert "FATAL" not in text.upper().replace("HALT_ON_ERROR", ""), [l for l in text.splitlines() if "FATAL" in l.upper()]
