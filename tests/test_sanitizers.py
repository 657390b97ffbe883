"""SURVEY.md section 5 (aux: sanitizers): the CPU oracle's known-answer suite and the golden vectors once more on the
AddressSanitizer + UndefinedBehaviorSanitizer build of the same sources (`make -C oracle asan`).  GPU sanitizers are not
available on this pool; the oracle is what every parity claim rests on, so it is the part that gets them."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    gcc = shutil.which("gcc")
    if not gcc:
        return None
    p = subprocess.run([gcc, "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_known_answers_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no gcc sanitizer runtimes")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    lib = os.path.join(ROOT, "oracle", "libldpc_oracle_asan.so")
    env = dict(os.environ, LDPC_ORACLE_LIB=lib, LD_PRELOAD=asan + ":" + ubsan, OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the reference's known answers, and two golden files per LLR family (f32 with NaN frames, i8, i32) for the decode loop
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_kats.py"),
                        os.path.join(ROOT, "tests", "test_goldens.py"), "-k",
                        "kats or (c_oracle and (TC128 or TM1280_i8 or TM2048_i32)) or (both_restatements and TC256_f32)"],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "runtime error" not in tail and "AddressSanitizer" not in tail, tail
