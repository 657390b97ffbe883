"""Condense tools/prof_frow.sh's output (gpurun_out/<round>/frow_*) into profiles/<round>/frow_summary.json and a table: per kernel and
code the mean duration (rocprofv3 kernel trace), algorithmic bytes per launch, their rate against the 8 TB/s HBM peak, wave-level VALU
instructions and HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, separate passes).    python tools/summarise_frow.py r04_final"""
import csv, glob, json, os, re, sys
name = sys.argv[1] if len(sys.argv) > 1 else "r04_final"
src, dst = f"gpurun_out/{name}", f"profiles/{name}"
os.makedirs(dst, exist_ok=True)
info = json.loads([l for l in open(f"{src}/frow_workload.json") if l.startswith("{")][-1])
# the workload runs TM8192 first, then TM2048: dispatches of one kernel name alternate by template argument / order; tell the codes apart by
# the kernel's template argument where it has one (encode / decode_bf), else by order (awgn: first 3 launches of a type = TM8192)
def code_of(kname, seq, per_code):
    m = re.search(r"<(?:ldpc::)?\(?(\d+)", kname)
    if "awgn" in kname:
        return "TM8192" if seq < per_code else "TM2048"
    if "encode_kernel_k4096" in kname: return "TM8192"
    if "encode_kernel<32>" in kname: return "TM2048"            # (32 = k / 32 words of data: k = 1024)
    m = re.search(r"kernel<(\d+)", kname)
    return {"8": "TM8192", "5": "TM2048"}.get(m.group(1) if m else "", "?")
def kind_of(kname):
    if "awgn_kernel<float>" in kname: return "awgn_f32"
    if "awgn_kernel<signed char>" in kname: return "awgn_i8"
    if "decode_bf_bs_kernel" in kname: return "decode_bf (bit-sliced)"
    if "decode_bf_kernel" in kname: return "decode_bf (byte per variable)"
    if "encode" in kname: return "encode"
    return None
rows = {}
for tag, label in (("frow_trace", ""), ("frow_trace_bytes", "")):
    files = sorted(glob.glob(f"{src}/{tag}/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1:]
    for f in files:
        seen = {}
        for r in csv.DictReader(open(f)):
            k = kind_of(r["Kernel_Name"])
            if not k: continue
            if tag == "frow_trace_bytes" and "decode_bf" not in k: continue
            if tag == "frow_trace" and k == "decode_bf (byte per variable)": continue
            seq = seen.get(k, 0); seen[k] = seq + 1
            code = code_of(r["Kernel_Name"], seq, 3)
            rows.setdefault((k, code), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pmc = {}
for i, counters in ((1, None), (2, None), (3, None)):
    for f in sorted(glob.glob(f"{src}/frow_pmc{i}/*/*_counter_collection.csv"), key=os.path.getmtime)[-1:]:
        seen = {}
        per = {}
        for r in csv.DictReader(open(f)):
            k = kind_of(r["Kernel_Name"])
            if not k: continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per.setdefault((k, r["Kernel_Name"], r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
            per[(k, r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        order = {}
        for (k, kname, disp), vals in sorted(per.items(), key=lambda x: int(x[0][2])):
            seq = order.get(k, 0); order[k] = seq + 1
            code = code_of(kname, seq, 3)
            for cn, v in vals.items():
                pmc.setdefault((k, code), {}).setdefault(cn, []).append(v)
out = {}
bytes_key = {"encode": "encode_bytes", "decode_bf (bit-sliced)": "decode_bf_bytes", "decode_bf (byte per variable)": "decode_bf_bytes",
             "awgn_f32": "awgn_f32_bytes", "awgn_i8": "awgn_i8_bytes"}
print(f"| kernel | code | frames | mean us (launches) | algorithmic GB/s | % of 8 TB/s | HBM bytes / algorithmic | VALU wave-instr per frame |")
print("|---|---|---|---|---|---|---|---|")
for (k, code), durs in sorted(rows.items()):
    if code not in info: continue
    d = durs[1:] if len(durs) > 1 else durs                      # the first launch of a kind is a warm-up
    us = sum(d) / len(d)
    alg = info[code][bytes_key[k]]
    p = pmc.get((k, code), {})
    mean = lambda xs: sum(xs) / len(xs) if xs else None
    hbm = (2 * mean(p.get("FETCH_SIZE", [])) + mean(p.get("WRITE_SIZE", []))) * 1024 if p.get("FETCH_SIZE") and p.get("WRITE_SIZE") else None
    valu = mean(p.get("SQ_INSTS_VALU", []))
    out[f"{k} {code}"] = {"frames": info[code]["frames"], "mean_us": us, "launches": len(d), "algorithmic_bytes": alg, "algorithmic_GBps": alg / us / 1e3,
                          "frac_of_hbm_peak": alg / us / 1e3 / 8000, "hbm_bytes_pmc": hbm, "valu_wave_insts": valu}
    print(f"| {k} | {code} | {info[code]['frames']} | {us:.1f} ({len(d)}) | {alg / us / 1e3:.0f} | {alg / us / 1e3 / 80:.2f} | "
          f"{hbm / alg if hbm else float('nan'):.2f} | {valu / info[code]['frames'] if valu else float('nan'):.0f} |")
json.dump(out, open(f"{dst}/frow_summary.json", "w"), indent=1)
