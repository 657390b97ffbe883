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


def test_host_side_of_the_library_under_tsan_with_sixteen_threads(tmp_path):
    """Round 5's review, item 5: the library's HOST code (capi.hip: argument checks, thread-local error string and staging, worker pool;
    encoder.cpp) built with -fsanitize=thread (`make -C labrador_ldpc_amd/csrc tsan`), driven by tests/c/threads_single_frame.c -- 16
    threads looping the reference-shaped single-frame symbols and the host helpers (copy_encode, hard_to_llrs) on thread-private
    buffers.  Without a GPU the decoders return `false` after the shared-state part of the call (device discovery, the error
    string); ThreadSanitizer must stay silent.  (With a GPU the un-instrumented HIP runtime is in the picture: skipped there; the
    plain GPU run is tests/test_c_boundary.py.)"""
    import labrador_ldpc_amd as la
    clang = "/opt/rocm/lib/llvm/bin/clang"
    rtdirs = [os.path.dirname(p) for p in __import__("glob").glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so")]
    if not os.path.exists(clang) or not rtdirs:
        pytest.skip("no clang ThreadSanitizer runtime")
    if la.device_count() > 0:
        pytest.skip("a GPU is present: TSan would watch the un-instrumented HIP runtime")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "labrador_ldpc_amd", "csrc"), "-s", "tsan"])
    tdir = os.path.join(ROOT, "build", "csrc_tsan")
    hipdir = os.path.dirname(la.HIP_RT_PATH) if getattr(la, "HIP_RT_PATH", None) else "/opt/rocm/lib"
    exe = str(tmp_path / "threads_tsan")
    cmd = [clang, "-std=c11", "-O1", "-g", "-Wall", "-Werror", "-pthread", "-fsanitize=thread", "-shared-libsan", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "threads_single_frame.c"), "-L" + tdir, "-llabrador_ldpc_hip", "-L" + hipdir, "-lamdhip64",
           "-Wl,-rpath," + tdir, "-Wl,-rpath," + hipdir, "-Wl,-rpath," + rtdirs[0], "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "16", "18"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 77 and "ThreadSanitizer" not in tail and "all returned false" in r.stdout, tail
