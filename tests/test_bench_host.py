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
    prof = {"library_build": "aaaaaaaaaaaaaaaa",
            "TM8192_f32": {"frames": 65536, "hbm_bytes_per_launch": 2.2e9, "valu_insts_per_launch": 6.0e9},
            "TC512_f32": {"frames": 65536, "hbm_bytes_per_launch": 1.4e8, "library_build": "bbbbbbbbbbbbbbbb"}}
    path = tmp_path / "hbm_traffic.json"
    path.write_text(json.dumps(prof))
    t, why = b.profile_counters("TM8192_f32", "aaaaaaaaaaaaaaaa", str(path))
    assert t and why is None
    t, why = b.profile_counters("TM8192_f32", "cccccccccccccccc", str(path))          # the library was rebuilt since the profile
    assert t is None and "stale profile" in why and "aaaaaaaaaaaaaaaa" in why and "cccccccccccccccc" in why
    t, why = b.profile_counters("TC512_f32", "aaaaaaaaaaaaaaaa", str(path))           # a per-kernel build id wins over the file's
    assert t is None and "stale" in why
    t, why = b.profile_counters("TM2048_f32", "aaaaaaaaaaaaaaaa", str(path))
    assert t is None and "no profile" in why
    t, why = b.profile_counters("TM8192_f32", "aaaaaaaaaaaaaaaa", str(tmp_path / "missing.json"))
    assert t is None and "no profile" in why


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
    t8192 = b.limiter_text(Code(8192, 1280, 30720), 4, 17.5)
    tc512 = b.limiter_text(Code(512, 64, 2048), 4, 15.0)
    assert "34053 B" in t8192 and "30720 edges" in t8192
    assert "2117 B" in tc512 and "2048 edges" in tc512 and "34053" not in tc512
    assert abs(b.VALU_PEAK_G - 1228.8) < 1e-6 and b.SHADER_CLOCK_UNDER_LOAD_GHZ == 2.30
