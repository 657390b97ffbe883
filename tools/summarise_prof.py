"""Collect the rocprofv3 outputs of tools/prof_final.sh (gpurun_out/<dir>) into profiles/<dir>:
kernel stats of the decode kernel, per-launch means of every PMC counter, corrected HBM traffic."""
import csv, glob, json, os, shutil, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r01_final"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01_final"
os.makedirs(dst, exist_ok=True)
for f in glob.glob(f"{src}/trace/*/*_kernel_stats.csv"):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "ldpc::" in r[0]]
    csv.writer(open(f"{dst}/kernel_stats_tm8192_f32.csv", "w", newline="")).writerows(keep)
summary = {}
for i in range(1, 6):
    for f in glob.glob(f"{src}/pmc{i}/*/*_counter_collection.csv"):
        shutil.copy(f, f"{dst}/pmc{i}_counters.csv")
        acc = {}
        for r in csv.DictReader(open(f)):
            if "decode_ms_" not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for name, per in acc.items():
            summary[name] = sum(per.values()) / len(per)          # mean per launch (65536 frames)
json.dump(summary, open(f"{dst}/pmc_summary_tm8192_f32_65536frames.json", "w"), indent=1)
for f in ("bench_default.json", "trace_bench.log"):
    if os.path.exists(f"{src}/{f}"):
        shutil.copy(f"{src}/{f}", f"{dst}/{'bench_under_rocprof.log' if f == 'trace_bench.log' else f}")
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    frames = 65536
    hbm = (2.0 * summary["FETCH_SIZE"] + summary["WRITE_SIZE"]) * 1024.0      # KB; FETCH_SIZE doubled: gfx950 correction
    json.dump({"TM8192_f32": {"frames": frames, "hbm_bytes_per_launch": hbm, "fetch_size_kb_raw": summary["FETCH_SIZE"],
                              "write_size_kb_raw": summary["WRITE_SIZE"],
                              "valu_insts_per_launch": summary.get("SQ_INSTS_VALU"), "lds_insts_per_launch": summary.get("SQ_INSTS_LDS"),
                              "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/prof_final.sh, {dst}); FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md; algorithmic bytes for 65536 frames = {frames * 34053}"}},
              open("profiles/hbm_traffic.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
