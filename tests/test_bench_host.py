"""Host-side logic of bench.py that needs no GPU: the committed profile's counters count for a bench line only if they were
collected on the library build the run loaded (round 2's review, item 6), and the job is one batch split into contiguous
shards whatever the number of GPUs."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_a_stale_profile_is_reported_as_stale(tmp_path):
    b = _bench()
    k8192, ktc, k2048 = b.profile_key("TM8192", "f32", 2.0, 25), b.profile_key("TC512", "f32", 2.0, 25), b.profile_key("TM2048", "f32", 2.0, 25)
    assert k8192 == "TM8192_f32_2dB_25it"
    prof = {"library_build": "aaaaaaaaaaaaaaaa",
            k8192: {"frames": 65536, "hbm_bytes_per_launch": 2.2e9, "valu_insts_per_launch": 6.0e9},
            ktc: {"frames": 65536, "hbm_bytes_per_launch": 1.4e8, "library_build": "bbbbbbbbbbbbbbbb"}}
    path = tmp_path / "hbm_traffic.json"
    path.write_text(json.dumps(prof))
    t, why = b.profile_counters(k8192, "aaaaaaaaaaaaaaaa", str(path))
    assert t and why is None
    t, why = b.profile_counters(k8192, "cccccccccccccccc", str(path))          # the library was rebuilt since the profile
    assert t is None and "stale profile" in why and "aaaaaaaaaaaaaaaa" in why and "cccccccccccccccc" in why
    t, why = b.profile_counters(ktc, "aaaaaaaaaaaaaaaa", str(path))           # a per-kernel build id wins over the file's
    assert t is None and "stale" in why
    t, why = b.profile_counters(k2048, "aaaaaaaaaaaaaaaa", str(path))
    assert t is None and "no profile" in why
    t, why = b.profile_counters(k8192, "aaaaaaaaaaaaaaaa", str(tmp_path / "missing.json"))
    assert t is None and "no profile" in why


def test_a_profile_counts_only_at_its_own_operating_point(tmp_path):
    """Round 3's review, weak #4: config 5's 2 dB entry carried a valu_issue derived from the 4 dB profile (9.4 passes per frame
    against 26).  Instruction counters are keyed by (code, type, Eb/N0, max_iters); a lookup at another operating point yields
    None with the reason, while HBM bytes per frame -- which do not depend on the operating point -- may still be used."""
    b = _bench()
    build = "aaaaaaaaaaaaaaaa"
    k4 = b.profile_key("TM5120", "i8", 4.0, 25)
    prof = {"library_build": build, k4: {"frames": 131072, "hbm_bytes_per_launch": 7.7e8, "valu_insts_per_launch": 4.3e9}}
    path = tmp_path / "hbm_traffic.json"
    path.write_text(json.dumps(prof))
    t, why = b.profile_counters(b.profile_key("TM5120", "i8", 4.0, 25), build, str(path))
    assert t and t["valu_insts_per_launch"] == 4.3e9
    for ebn0, maxit in ((2.0, 25), (4.0, 50), (4.5, 25)):
        t, why = b.profile_counters(b.profile_key("TM5120", "i8", ebn0, maxit), build, str(path))
        assert t is None and "no profile" in why and k4 in why, (ebn0, maxit, why)          # ... and it says which point WAS profiled
    t, key, why = b.traffic_profile("TM5120", "i8", build, str(path))
    assert t and key == k4 and why is None
    t, key, why = b.traffic_profile("TM5120", "i8", "cccccccccccccccc", str(path))
    assert t is None and "stale" in why
    t, key, why = b.traffic_profile("TM5120", "f32", build, str(path))            # another type's kernel is another kernel
    assert t is None and "no profile" in why


def test_roofline_of_prints_null_valu_issue_for_an_unprofiled_operating_point(tmp_path, monkeypatch):
    b = _bench()
    build = "aaaaaaaaaaaaaaaa"
    prof = {"library_build": build,
            b.profile_key("TM2048", "i8", 4.0, 25): {"frames": 131072, "hbm_bytes_per_launch": 3.1e8, "valu_insts_per_launch": 4.3e9},
            b.profile_key("TM5120", "i8", 4.0, 25): {"frames": 131072, "hbm_bytes_per_launch": 8.5e9, "valu_insts_per_launch": 2.3e9}}
    path = tmp_path / "profiles" / "hbm_traffic.json"
    path.parent.mkdir()
    path.write_text(json.dumps(prof))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))

    class Code:
        def __init__(self, n, out, e): self._n, self._o, self._e = n, out, e
        def n(self): return self._n
        def output_len(self): return self._o
        def paritycheck_sum(self): return self._e
    tm2048, tm5120 = Code(2048, 320, 7680), Code(5120, 704, 19968)
    roof, valu = b.roofline_of(tm2048, "TM2048", "i8", 0, 524288, 23.5, 8.4, build, 4.0, 25)
    assert valu and abs(valu["achieved"] - 4.3e9 * 4 / 23.5e-3 / 1e9) < 1e-6 and roof["traffic"] == 3.1e8 * 4
    roof, valu = b.roofline_of(tm2048, "TM2048", "i8", 0, 524288, 61.2, 25.0, build, 2.0, 25)
    assert valu is None and "no profile of TM2048_i8_2dB_25it" in roof["valu_issue_note"]
    assert roof["traffic"] == 3.1e8 * 4 and "TM2048_i8_4dB_25it" in roof["traffic_source"]      # bytes per frame: any operating point
    # the rate-4/5 codes' default: two waves per codeword group, LLR planes in LDS -- bytes per frame like every other kernel
    roof, valu = b.roofline_of(tm5120, "TM5120", "i8", 0, 524288, 17.0, 8.4, build, 4.0, 25)
    assert roof["kernel"] == "decode_ms_bs_split_kernel" and roof["traffic"] == 8.5e9 * 4 and "Infinity Cache" not in roof["traffic_source"] and valu
    roof, valu = b.roofline_of(tm5120, "TM5120", "i8", 0, 524288, 37.6, 25.0, build, 2.0, 25)
    assert roof["traffic"] == 8.5e9 * 4 and valu is None
    # a batch below the dispatch threshold runs the f32-pipe kernel and is labelled so (round 4 advice): the library is asked
    roof, _ = b.roofline_of(tm5120, "TM5120", "i8", 0, 1024, 0.1, 8.4, build, 4.0, 25)
    assert roof["kernel"] == "decode_ms_kernel"
    roof, _ = b.roofline_of(tm5120, "TM5120", "i8", 64, 1024, 0.1, 8.4, build, 4.0, 25)
    assert roof["kernel"] == "decode_ms_bs_split_kernel"
    assert b.kernel_name("TM8192", "i8", 0, 4096) == "decode_ms_bs_kernel" and b.kernel_name("TM8192", "i8", 0, 512) == "decode_ms_pair_kernel"
    assert b.kernel_name("TM8192", "i8", 256, 4096) == "decode_ms_pair_kernel"          # a launch flag keeps the f32-pipe kernel
    assert b.kernel_name("TC512", "i8", 0, 1 << 20) == "decode_ms_kernel"
    # slot refill (TM1536, TM1280): by name at any size, by default from 65 536 frames up, never with the STATIC flag
    for code, lock, refill in (("TM1536", "decode_ms_bs_kernel", "decode_ms_bs_refill_kernel"), ("TM1280", "decode_ms_bs_split_kernel", "decode_ms_bs_split_refill_kernel")):
        assert b.kernel_name(code, "i8", 0, 65536) == refill and b.kernel_name(code, "i8", 0, 65535) == lock and b.kernel_name(code, "i8", 0, 1024) == "decode_ms_kernel"
        assert b.kernel_name(code, "i8", 64, 8) == refill and b.kernel_name(code, "i8", 64 | 256, 1 << 20) == lock
    assert b.kernel_name("TM5120", "i8", 0, 1 << 20) == "decode_ms_bs_split_kernel" and b.kernel_name("TM2048", "i8", 64, 1 << 20) == "decode_ms_bs_kernel"


def test_an_alternative_library_build_never_gets_profile_figures(tmp_path, monkeypatch):
    """tools/bs_alt_build.sh links experimental kernels with the product's capi.o, so such a library reports the product's build id
    (round 4 advice): with LABRADOR_LDPC_HIP_LIB pointing anywhere but at the in-tree library no committed profile is attached."""
    b = _bench()
    build = "aaaaaaaaaaaaaaaa"
    key = b.profile_key("TM8192", "f32", 2.0, 25)
    path = tmp_path / "hbm_traffic.json"
    path.write_text(json.dumps({"library_build": build, key: {"frames": 65536, "hbm_bytes_per_launch": 2.2e9, "valu_insts_per_launch": 5.5e9}}))
    monkeypatch.delenv("LABRADOR_LDPC_HIP_LIB", raising=False)
    assert b.profile_counters(key, build, str(path))[0] and b.traffic_profile("TM8192", "f32", build, str(path))[0]
    monkeypatch.setenv("LABRADOR_LDPC_HIP_LIB", os.path.join(ROOT, "labrador_ldpc_amd", "liblabrador_ldpc_hip.so"))
    assert b.profile_counters(key, build, str(path))[0]                                  # the in-tree library by its own path: fine
    monkeypatch.setenv("LABRADOR_LDPC_HIP_LIB", str(tmp_path / "liblabrador_ldpc_hip_experiment.so"))
    t, why = b.profile_counters(key, build, str(path))
    assert t is None and "alternative library" in why
    t, _, why = b.traffic_profile("TM8192", "f32", build, str(path))
    assert t is None and "alternative library" in why


def test_the_committed_profile_names_the_build_it_was_collected_on():
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
        doc = json.load(f)
    entries = {k: v for k, v in doc.items() if isinstance(v, dict) and v.get("frames")}
    assert entries
    for key, t in entries.items():
        assert t.get("library_build") or doc.get("library_build"), f"{key}: no library_build recorded"


def test_roofline_fields_and_limiter_text_are_per_configuration():
    b = _bench()

    class Code:                                           # the three numbers the text needs
        def __init__(self, n, out, e): self._n, self._o, self._e = n, out, e
        def n(self): return self._n
        def output_len(self): return self._o
        def paritycheck_sum(self): return self._e
    t8192 = b.limiter_text(Code(8192, 1280, 30720), 4, 18.5)
    tc512 = b.limiter_text(Code(512, 64, 2048), 4, 15.0)
    assert "34053 B" in t8192 and "30720 edges" in t8192 and "x 18.5 iterations executed" in t8192
    # a batch in which nothing converges executed max_iters passes per frame, not max_iters + 1 (round 4 review, weak #4)
    assert "x 25.0 iterations executed" in b.limiter_text(Code(5120, 704, 19968), 1, 25.0)
    assert "2117 B" in tc512 and "2048 edges" in tc512 and "34053" not in tc512
    assert abs(b.VALU_PEAK_G - 1228.8) < 1e-6 and b.SHADER_CLOCK_UNDER_LOAD_GHZ == 2.30


def test_build_id_is_a_function_of_the_sources_not_of_the_produced_bytes(tmp_path):
    """Round 3's review, weak #5: the profile staleness guard was keyed on sha256(.so), which hipcc does not reproduce.  The id is
    now a hash of the sources, the public header, the flags and the compiler version (csrc/build_id.sh, baked into the library as
    labrador_ldpc_hip_build_id()): two evaluations of one tree agree, a one-character kernel edit changes it, a flag edit too."""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "labrador_ldpc_amd", "csrc")
    hdr = os.path.join(ROOT, "include", "labrador_ldpc_hip.h")
    hipcc = "/opt/rocm/bin/hipcc"

    def ident(d, h=hdr, flags="gfx950 -O3"):
        return subprocess.run(["sh", os.path.join(csrc, "build_id.sh"), d, h, flags, hipcc], capture_output=True, text=True, check=True).stdout.strip()
    copy = tmp_path / "csrc"
    shutil.copytree(csrc, copy, ignore=shutil.ignore_patterns("*.o", "profiles"))
    a, b_, c = ident(csrc), ident(csrc), ident(str(copy))
    assert len(a) == 16 and int(a, 16) >= 0 and a == b_ == c
    kernel = copy / "decode_ms_kernel.hpp"
    text = kernel.read_text()
    kernel.write_text(text.replace("v_min3_f32", "v_min3_f33", 1))
    assert text != kernel.read_text() and ident(str(copy)) != a                     # one character of one kernel
    kernel.write_text(text)
    assert ident(str(copy)) == a
    assert ident(str(copy), flags="gfx950 -O2") != a


def test_the_loaded_library_reports_the_build_id_of_this_tree():
    import subprocess
    import labrador_ldpc_amd as la
    want = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "labrador_ldpc_amd", "csrc"), "print-build-id"],
                          capture_output=True, text=True, check=True).stdout.strip()
    got = la.lib.labrador_ldpc_hip_build_id().decode()
    assert len(got) == 16 and got == want, "the .so is older than the sources: run make -C labrador_ldpc_amd/csrc"
    b = _bench()
    assert b.library_build_id() == got


def test_lds_bytes_per_pass_follows_the_block_structure():
    """lds_bytes_per_s of the bench line (hard part 6): TM8192 has 8 exchanged blocks of 2048 edges in 3
    block columns -> (8 x 3 + 3) x 2048 elements of 4 bytes per pass."""
    import labrador_ldpc_amd as la
    b = _bench()
    assert b.lds_bytes_per_pass(la.LDPCCode.TM8192) == (8 * 3 + 3) * 2048 * 4
    assert b.lds_bytes_per_pass(la.LDPCCode.TM2048) == (8 * 3 + 3) * 512 * 4


def test_job_digest_is_split_invariant():
    """The digest of the bench line sums per-frame terms weighted by the GLOBAL frame index: shards add up to the whole."""
    import torch
    b = _bench()

    class W(b.Workload):
        def __init__(self, first, out, iters, succ):
            self.first_frame, self.frames, self.out, self.iters, self.succ = first, out.shape[0], out, iters, succ
    g = torch.Generator().manual_seed(5)
    out = torch.randint(0, 256, (1000, 64), dtype=torch.uint8, generator=g)
    iters = torch.randint(0, 26, (1000,), dtype=torch.int32, generator=g)
    succ = (iters < 25).to(torch.uint8)
    whole = W(0, out, iters, succ).sums()
    parts = [W(lo, out[lo:hi], iters[lo:hi], succ[lo:hi]).sums() for lo, hi in ((0, 333), (333, 334), (334, 1000))]
    assert whole[0] == sum(p[0] for p in parts) and whole[1] == sum(p[1] for p in parts)
    assert whole[2] == sum(p[2] for p in parts) % b.DIGEST_MOD
    flipped = out.clone()
    flipped[777, 3] ^= 1
    assert W(0, flipped, iters, succ).sums()[2] != whole[2]
    swapped = out.clone()
    swapped[[10, 11]] = swapped[[11, 10]]
    assert W(0, swapped, iters, succ).sums()[2] != whole[2] or bool((out[10] == out[11]).all())


def test_whole_job_check_against_the_committed_answers():
    """Round 5's review, next #1: every bench line compares the job's exact results with the committed one-GPU answers."""
    b = _bench()
    key = b.job_key("TM8192", "f32", 4194304, 2.0, 25, 256)
    assert key == "TM8192_f32_4194304f_2dB_25it_pool256" and key in b.EXPECTED_JOBS          # the headline job is committed ...
    assert b.job_key("TM8192", "i8", 4194304, 2.0, 25, 256) in b.EXPECTED_JOBS                # ... and the i8 job
    want = b.EXPECTED_JOBS[key]
    ok = b.whole_job_check(key, want["iters_sum"], want["failed_frames"], want["job_digest"])
    assert ok["match"] is True and "got" not in ok
    for field, bad in (("iters_sum", want["iters_sum"] + 1), ("failed_frames", want["failed_frames"] - 1), ("job_digest", "00000000000000")):
        got = dict(want, **{field: bad})
        r = b.whole_job_check(key, got["iters_sum"], got["failed_frames"], got["job_digest"])
        assert r["match"] is False and r["got"][field] == bad and r["expected"][field] == want[field]
    r = b.whole_job_check(b.job_key("TM8192", "f32", 12345, 2.0, 25, 256), 1, 2, "03")
    assert r["match"] is None and "no committed answer" in r["note"]                          # an unknown job is neither a pass nor a failure
    # another pool, another cap, another operating point: another job
    assert len({b.job_key("TM8192", "f32", 4194304, e, m, p) for e in (2.0, 2.5) for m in (25, 50) for p in (256, 16)}) == 8
    for k, v in b.EXPECTED_JOBS.items():
        assert set(v) == {"iters_sum", "failed_frames", "job_digest"} and len(v["job_digest"]) == 14 and int(v["job_digest"], 16) < b.DIGEST_MOD, k
