#!/usr/bin/env python3
"""Record the whole-job known answers bench.py carries (EXPECTED_JOBS): every listed job is decoded on ONE GPU by bench.py itself
and its exact results -- iters_sum, failed_frames, job_digest -- are written as the dict literal to paste into bench.py.
   gpurun -- python3 tools/record_expected_jobs.py [--check]     -> gpurun_out/expected_jobs.{json,py.txt}
--check: compare with what bench.py holds instead (non-zero exit on a difference): the answers must not depend on the box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (code, dtype, total frames, Eb/N0): the headline job, the same job with i8 LLRs, the other BASELINE configs as bench.py's
# `configs` runs them, and the small jobs of tests/test_gpu_multirank.py
JOBS = [("TM8192", "f32", 4194304, 2.0), ("TM8192", "i8", 4194304, 2.0),
        ("TC512", "f32", 65536, 2.0), ("TM2048", "f32", 1048576, 2.0),
        ("TM5120", "i8", 524288, 4.0), ("TM5120", "i8", 524288, 2.0), ("TM8192", "i8", 524288, 2.0),
        ("TM5120", "i8", 4194304, 4.0), ("TM5120", "i8", 4194304, 2.0),
        ("TM8192", "f32", 65536, 2.0), ("TM8192", "f32", 32769, 2.0), ("TM8192", "f32", 16384, 2.0), ("TM8192", "f32", 40961, 2.0),
        ("TM8192", "i8", 40961, 2.0), ("TM5120", "i8", 40961, 4.0)]


def main():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = {}
    for code, dtype, total, ebn0 in JOBS:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--code", code, "--dtype", dtype, "--total-frames", str(total),
                            "--ebn0", str(ebn0), "--steps", "1", "--warmup", "0", "--no-cpu", "--no-configs"], capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(f"{code} {dtype} {total}: no line (rc {r.returncode}): {r.stderr[-500:]}", file=sys.stderr)
            continue
        d = json.loads(lines[-1])
        key = bench.job_key(code, dtype, total, ebn0, 25, 256)
        out[key] = {"iters_sum": d["diag"]["iters_sum"], "failed_frames": d["diag"]["failed_frames"], "job_digest": d["diag"]["job_digest"]}
        print(key, out[key], "rc", r.returncode, flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "expected_jobs.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "gpurun_out", "expected_jobs.py.txt"), "w") as f:
        f.write("EXPECTED_JOBS = {\n")
        for k, v in out.items():
            f.write(f'    "{k}": {{"iters_sum": {v["iters_sum"]}, "failed_frames": {v["failed_frames"]}, "job_digest": "{v["job_digest"]}"}},\n')
        f.write("}\n")
    if "--check" in sys.argv:
        bad = {k: (v, bench.EXPECTED_JOBS.get(k)) for k, v in out.items() if bench.EXPECTED_JOBS.get(k) != v}
        print("differences from bench.EXPECTED_JOBS:", bad or "none")
        return 1 if bad else 0
    return 0


if __name__ == "__main__":
    sys.exit(main())
