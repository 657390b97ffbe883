"""Condense gpurun_out/<round> (tools/prof_round.sh) into profiles/<round> and refresh profiles/hbm_traffic.json:
kernel-trace stats of every decode kernel of the default bench command, per-launch means of every PMC counter per
configuration, HBM traffic corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2 on gfx950) + WRITE_SIZE.
    python tools/summarise_round.py r03_final"""
import csv, glob, json, os, shutil, sys
name = sys.argv[1] if len(sys.argv) > 1 else "r03_final"
src, dst = f"gpurun_out/{name}", f"profiles/{name}"
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    """gpurun_out/ accumulates over calls: keep only the most recent file of each kind"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


for f in newest(f"{src}/trace/*/*_kernel_stats.csv"):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "ldpc::" in r[0]]
    csv.writer(open(f"{dst}/kernel_stats_bench_default.csv", "w", newline="")).writerows(keep)
for f in ("bench_default.json", "trace_bench.log"):
    if os.path.exists(f"{src}/{f}"):
        shutil.copy(f"{src}/{f}", f"{dst}/{'bench_under_rocprof.log' if f == 'trace_bench.log' else f}")
# tag of tools/prof_round.sh -> (frames per launch, algorithmic bytes per frame, Eb/N0, max_iters): a profile entry belongs to ONE
# operating point (bench.py profile_key: "<code>_<dtype>_<Eb/N0>dB_<max_iters>it")
CONFIGS = {"TM8192_f32": (65536, 8192 * 4 + 1280 + 5, 2.0, 25), "TC512_f32": (65536, 512 * 4 + 64 + 5, 2.0, 25),
           "TM2048_f32": (262144, 2048 * 4 + 320 + 5, 2.0, 25), "TM5120_i8": (131072, 5120 + 704 + 5, 4.0, 25),
           "TM5120_i8_2dB": (131072, 5120 + 704 + 5, 2.0, 25), "TM8192_i8": (65536, 8192 + 1280 + 5, 2.0, 25)}
traffic, allsum = {}, {}
for tag, (frames, alg, ebn0, maxit) in CONFIGS.items():
    key = f"{'_'.join(tag.split('_')[:2])}_{ebn0:g}dB_{maxit}it"
    summary = {}
    for i in range(1, 6):
        for f in newest(f"{src}/{tag}.pmc{i}/*/*_counter_collection.csv"):
            if tag == "TM8192_f32":
                shutil.copy(f, f"{dst}/pmc{i}_counters_{tag}.csv")
            acc = {}
            for r in csv.DictReader(open(f)):
                if "decode_ms_" not in r["Kernel_Name"]:
                    continue
                acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            for cname, per in acc.items():
                summary[cname] = sum(per.values()) / len(per)          # mean per launch
    if not summary:
        continue
    allsum[tag] = dict(summary, frames_per_launch=frames, profile_key=key)
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        hbm = (2.0 * summary["FETCH_SIZE"] + summary["WRITE_SIZE"]) * 1024.0      # KB -> B; FETCH_SIZE doubled: gfx950 correction
        traffic[key] = {"frames": frames, "ebn0_db": ebn0, "max_iters": maxit, "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": frames * alg,
                        "hbm_over_algorithmic": hbm / (frames * alg),
                        "fetch_size_kb_raw": summary["FETCH_SIZE"], "write_size_kb_raw": summary["WRITE_SIZE"],
                        "valu_insts_per_launch": summary.get("SQ_INSTS_VALU"), "lds_insts_per_launch": summary.get("SQ_INSTS_LDS"),
                        "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/prof_round.sh, {dst}); "
                                "FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md"}
json.dump(allsum, open(f"{dst}/pmc_summary.json", "w"), indent=1)
mix = f"{dst}/valu_mix_tm8192_f32.json"              # tools/valu_mix.py (static instruction mix of the iteration loop)
if os.path.exists(mix) and "TM8192_f32_2dB_25it" in traffic:
    traffic["TM8192_f32_2dB_25it"]["avg_issue_cycles_per_instruction"] = json.load(open(mix))["avg_issue_cycles_per_instruction"]
# the library build the counters were collected on (the bench line of the same gpurun call names it): bench.py reports
# `traffic` / `valu_issue` only when the library it loaded is this one (tests/test_bench_host.py)
build = None
try:
    with open(f"{src}/bench_default.json") as fh:
        build = json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])["config"]["library_build"]
except Exception as e:
    print("no library_build in", f"{src}/bench_default.json:", e)
if traffic:
    for t in traffic.values():
        t["library_build"] = build
    traffic["library_build"] = build
    json.dump(traffic, open("profiles/hbm_traffic.json", "w"), indent=1)
for tag, s in allsum.items():
    w = s.get("SQ_WAVE_CYCLES")
    if w:
        print(tag, {k: round(s[k] / w, 3) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS") if k in s},
              "LDS conflict share", round(s.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, s.get("SQ_LDS_IDX_ACTIVE", 1)), 3),
              "HBM/alg", round(traffic[s["profile_key"]]["hbm_over_algorithmic"], 4) if s["profile_key"] in traffic else None)
