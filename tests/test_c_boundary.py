"""The boundary as a C compiler sees it: the header (include/labrador_ldpc_hip.h, and its
source-compat name include/labrador_ldpc.h) against the shared library, through real C translation
units -- not ctypes with hand-typed argtypes.

CPU tests: tests/c/example_smoke.c (this repo's C client, shaped like the reference's
capi/examples/example.c:23-95) compiles for every code with -Wall -Werror, links against the .so,
and its host-side half runs (static macro sizes == run-time size functions, encoder, LLR helpers);
the reference's spellings (_TM6140 aliases, one-argument LABRADOR_LDPC_N_() helpers) compile; and,
where /root/reference exists (this container only), the reference's OWN example.c compiles and
links against our header and library unchanged.
GPU test: the same client runs end to end (decode_ms_f32 200 iterations, decode_bf) for every code.
"""
import os
import shutil
import subprocess

import pytest

import labrador_ldpc_amd as la

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
SRC = os.path.join(ROOT, "tests", "c", "example_smoke.c")
NAMES = ["TC128", "TC256", "TC512", "TM1280", "TM1536", "TM2048", "TM5120", "TM6144", "TM8192"]
CC = shutil.which("cc") or shutil.which("gcc")
REF_EXAMPLE = "/root/reference/capi/examples/example.c"

pytestmark = pytest.mark.skipif(CC is None, reason="no C compiler")


def hip_libdir():
    for d in ([os.path.dirname(la.HIP_RT_PATH)] if getattr(la, "HIP_RT_PATH", None) else []) + ["/opt/rocm/lib"]:
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    pytest.skip("libamdhip64.so not found")


def link_args():
    libdir, hipdir = os.path.dirname(la.LIB_PATH), hip_libdir()
    return ["-L" + libdir, "-llabrador_ldpc_hip", "-L" + hipdir, "-lamdhip64",
            "-Wl,-rpath," + libdir, "-Wl,-rpath," + hipdir]


def build_client(tmp_path, name, src=SRC, extra=()):
    exe = str(tmp_path / f"client_{name}")
    cmd = [CC, "-std=c11", "-O1", "-Wall", "-Werror", "-I" + INC, f"-DCODE={name}", *extra, src, "-o", exe, *link_args()]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stderr}"
    return exe


@pytest.mark.parametrize("name", NAMES)
def test_c_client_compiles_links_and_host_half_runs(tmp_path, name):
    exe = build_client(tmp_path, name)
    r = subprocess.run([exe, "--host-only"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, r.stdout + r.stderr
    assert "host checks ok" in r.stdout


def test_reference_spellings_compile(tmp_path):
    """labrador_ldpc.h:42-115 spellings a reference-era source may use, incl. the TM6140 quirk."""
    src = tmp_path / "spellings.c"
    src.write_text(r'''
#include "labrador_ldpc.h"
#define MYCODE TM6144
_Static_assert(LABRADOR_LDPC_N(MYCODE) == 6144, "n of TM6144 (the reference header says 6140)");
_Static_assert(LABRADOR_LDPC_MS_WORKING_LEN(MYCODE) == LABRADOR_LDPC_MS_WORKING_LEN_TM6140, "alias");
_Static_assert(LABRADOR_LDPC_BF_WORKING_LEN_TM6140 == 7168, "bf");
_Static_assert(LABRADOR_LDPC_MS_WORKING_LEN_TM6140 == 60416, "ms");
_Static_assert(LABRADOR_LDPC_MS_WORKING_U8_LEN_TM6140 == 384, "u8");
_Static_assert(LABRADOR_LDPC_OUTPUT_LEN_TM6140 == 896, "out");
_Static_assert(LABRADOR_LDPC_N_(TC512) == 512 && LABRADOR_LDPC_K_(TC512) == 256, "one-argument helpers");
_Static_assert(LABRADOR_LDPC_CODE_(TM8192) == 8 && LABRADOR_LDPC_CODE(MYCODE) == LABRADOR_LDPC_CODE_TM6144, "enum");
_Static_assert(LABRADOR_LDPC_OUTPUT_LEN_(TM1280) == 176 && LABRADOR_LDPC_MS_WORKING_U8_LEN_(TM1280) == 48
               && LABRADOR_LDPC_MS_WORKING_LEN_(TM1280) == 12160 && LABRADOR_LDPC_BF_WORKING_LEN_(TM1280) == 1408, "helpers");
int main(void) { return 0; }
''')
    r = subprocess.run([CC, "-std=c11", "-Wall", "-Werror", "-I" + INC, "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_header_is_valid_cxx_and_c99(tmp_path):
    src = tmp_path / "inc.c"
    src.write_text('#include "labrador_ldpc_hip.h"\nint main(void) { return (int)sizeof(struct labrador_ldpc_hip_opts) == 0; }\n')
    for std, lang in (("-std=c99", "c"), ("-std=c++17", "c++")):
        r = subprocess.run([CC, std, "-x", lang, "-Wall", "-Werror", "-pedantic", "-I" + INC, "-fsyntax-only", str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_every_declared_function_links(tmp_path):
    """One translation unit that takes the address of every function the header declares: the
    header and the .so agree on the names at link time (ctypes never proves that)."""
    import re
    hdr = open(os.path.join(INC, "labrador_ldpc_hip.h")).read()
    names = sorted(set(re.findall(r"\b(labrador_ldpc_[a-z0-9_]+)\s*\(", hdr)))
    src = tmp_path / "addr.c"
    src.write_text('#include "labrador_ldpc.h"\n#include <stdio.h>\nint main(void) {\n  void *p[] = {'
                   + ", ".join(f"(void *)(size_t){n}" for n in names)
                   + '};\n  for (unsigned i = 0; i < sizeof p / sizeof p[0]; i++) if (!p[i]) return 1;\n'
                   + '  printf("%u\\n", (unsigned)(sizeof p / sizeof p[0]));\n  return 0;\n}\n')
    exe = str(tmp_path / "addr")
    r = subprocess.run([CC, "-std=c11", "-I" + INC, str(src), "-o", exe, *link_args()], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and int(r.stdout) == len(names) >= 30


@pytest.mark.skipif(not os.path.exists(REF_EXAMPLE), reason="reference tree not present (GPU box)")
def test_reference_example_compiles_and_links_unchanged(tmp_path):
    """The reference's only C client, read where it lies, against OUR include directory and library."""
    exe = str(tmp_path / "ref_example")
    r = subprocess.run([CC, "-I" + INC, REF_EXAMPLE, "-o", exe, *link_args()], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


BATCH_SRC = os.path.join(ROOT, "tests", "c", "batch_smoke.c")


def test_batched_c_client_compiles_and_reports_no_device(tmp_path):
    """The batched half of the header from C: compiles with -Wall -Werror, links, and without a GPU exits 77."""
    exe = build_client(tmp_path, "TM2048", src=BATCH_SRC)
    if la.device_count() > 0:
        pytest.skip("a GPU is present: the run is covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("frames", [1, 3001, 40000])
def test_batched_c_client_on_gpu(tmp_path, frames):
    """encode_batch -> decode_ms_batch_i8 on one device, on DEVICE_ALL and on a device list, from C, host buffers."""
    exe = build_client(tmp_path, "TM2048", src=BATCH_SRC)
    r = subprocess.run([exe, str(frames)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok:")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_c_client_end_to_end_on_gpu(tmp_path, name):
    exe = build_client(tmp_path, name)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok:")


LOOP_SRC = os.path.join(ROOT, "tests", "c", "device_loop.c")
HIP_INC = ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"), reason="no HIP headers")
def test_device_loop_client_compiles_and_reports_no_device(tmp_path):
    """One host thread looping the MEM_DEVICE calls over every device: compiles as C with -Wall -Werror, links; no GPU -> 77."""
    exe = build_client(tmp_path, "TM2048", src=LOOP_SRC, extra=HIP_INC)
    if la.device_count() > 0:
        pytest.skip("a GPU is present: the run is covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name,frames,parts", [("TM2048", 20011, 0), ("TM2048", 20011, 4), ("TM8192", 4099, 3), ("TC512", 7, 8)])
def test_device_resident_shards_from_one_thread_equal_the_one_call_job(tmp_path, name, frames, parts):
    """labrador_ldpc_decode_ms_batch_f32(MEM_DEVICE) looped over the devices (or, with `parts`, over that many shards cycling over
    them: four streams on one GPU) from ONE thread: every shard, generated by global frame index on its own device and decoded on
    its own stream, equals its slice of the job decoded by one call (round 3's review, weak #7 ii; shards for real with > 1 GPU)."""
    exe = build_client(tmp_path, name, src=LOOP_SRC, extra=HIP_INC)
    r = subprocess.run([exe, str(frames)] + ([str(parts)] if parts else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok:")


MULTI_SRC = os.path.join(ROOT, "tests", "c", "multi_call.c")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"), reason="no HIP headers")
def test_multi_call_client_compiles_and_reports_no_device(tmp_path):
    """labrador_ldpc_decode_ms_batch_*_multi as a C compiler sees it (pointer-to-pointer arguments, const placement): -Wall -Werror."""
    exe = build_client(tmp_path, "TM2048", src=MULTI_SRC, extra=HIP_INC)
    if la.device_count() > 0:
        pytest.skip("a GPU is present: the run is covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name,frames,parts", [("TM2048", 20011, 0), ("TM2048", 20011, 3), ("TM8192", 4099, 3), ("TM5120", 9001, 2), ("TC512", 7, 8)])
def test_device_resident_parts_through_one_multi_call_equal_the_one_call_job(tmp_path, name, frames, parts):
    """Round 4's review, item 7: per-device device-resident buffers handed to ONE call (labrador_ldpc_decode_ms_batch_{f32,i8}_multi),
    decoded by the library's per-device workers -- every part equals its slice of the job decoded by one call; with one GPU `parts`
    entries all name device 0 (three workers, three streams); shards for real with more GPUs.  Also: an empty part, a bad ordinal."""
    exe = build_client(tmp_path, name, src=MULTI_SRC, extra=HIP_INC)
    r = subprocess.run([exe, str(frames)] + ([str(parts)] if parts else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok:") == 2


THREADS_SRC = os.path.join(ROOT, "tests", "c", "threads_single_frame.c")


def test_threads_client_compiles_and_reports_no_device(tmp_path):
    """The unchanged-perftest shape (perftest/src/main.rs:39-45: every worker loops single-frame decodes on buffers of its own)
    from C with -Wall -Werror -pthread; without a GPU every call says `false` identically from 16 threads: exit 77."""
    exe = build_client(tmp_path, "TM2048", src=THREADS_SRC, extra=("-pthread",))
    if la.device_count() > 0:
        pytest.skip("a GPU is present: the run is covered by the gpu test")
    r = subprocess.run([exe, "16", "12"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 77, r.stdout + r.stderr
    assert "all returned false" in r.stdout and "DIFFERENTLY" not in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("threads,trials", [(16, 54), (3, 90)])
def test_n_threads_on_the_reference_shaped_single_frame_symbols(tmp_path, threads, trials):
    """Round 5's review, missing #3: the reference guarantees re-entrancy (src/lib.rs:15-17) and its harness relies on it
    (perftest/src/main.rs:39-45).  16 host threads loop labrador_ldpc_decode_ms_{f32,i8,i16,f64} / labrador_ldpc_decode_bf over
    mixed codes on caller-owned, thread-private buffers; every result (flag, iteration count, every output byte) must equal the
    single-threaded pass over the same trials."""
    exe = build_client(tmp_path, "TM2048", src=THREADS_SRC, extra=("-pthread",))
    r = subprocess.run([exe, str(threads), str(trials)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok:")
    print(r.stdout)
