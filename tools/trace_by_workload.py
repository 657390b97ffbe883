"""Per-dispatch durations of the decode kernels in the rocprofv3 --kernel-trace of the default bench command
(tools/prof_round.sh), grouped by kernel, in launch order -> profiles/<round>/kernel_trace_by_workload.txt
    python tools/trace_by_workload.py r03_final"""
import csv, glob, json, os, re, subprocess, sys
name = sys.argv[1] if len(sys.argv) > 1 else "r03_final"
src, dst = f"gpurun_out/{name}", f"profiles/{name}"
trace = sorted(glob.glob(f"{src}/trace/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
build = "?"
try:
    with open(f"{src}/bench_default.json") as fh:
        build = json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])["config"]["library_build"]
except Exception:
    pass
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
by = {}
for r in rows:
    k = r["Kernel_Name"]
    if "decode_ms" not in k:
        continue
    k = re.sub(r"^void ldpc::", "", k).split("(")[0]
    by.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = [f"rocprofv3 --kernel-trace of `python bench.py --no-cpu` (the default command; --steps 5 --warmup 1) on library build {build},",
       "per-dispatch durations in launch order, ms:", ""]
for k, d in by.items():
    out.append(f"{k}: {len(d)} launches")
    shown = d if len(d) <= 24 else d[:24]
    out.append("   " + ", ".join(f"{x:.3f}" for x in shown) + (" ..." if len(d) > 24 else ""))
    if "pair_kernel<8, float" in k:
        big = [x for x in d if x > 200]
        small = [x for x in d if x <= 200]
        if len(big) >= 6:
            m = sum(big[-5:]) / 5
            out.append(f"   whole config 4 (4 194 304 frames): {len(big)} launches (1 warm-up + 5 timed), timed five: mean {m:.2f} ms -> {4194304 / m / 1e3:.3f} M codewords/s")
        if small:
            m = sum(small) / len(small)
            out.append(f"   524 288-frame slices (first, last; 1 warm-up + 5 timed each): mean {m:.2f} ms -> {524288 / m / 1e3:.3f} M codewords/s")
    elif len(d) > 100:
        out.append(f"   mean of the last 100: {sum(d[-100:]) / 100:.4f} ms; first five: {sum(d[:5]) / 5:.4f} ms")
    else:
        out.append(f"   mean {sum(d) / len(d):.3f} ms")
open(f"{dst}/kernel_trace_by_workload.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
