"""FETCH_SIZE of the bit-sliced decode kernel in one rocprofv3 --pmc output directory (tools/nt_experiment.sh)."""
import csv, glob, sys
tot = {}
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "decode_ms_bs_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot.setdefault(r["Dispatch_Id"], 0.0)
            tot[r["Dispatch_Id"]] += float(r["Counter_Value"])
if tot:
    v = sorted(tot.values())
    m = v[len(v) // 2]
    print(f"   Eb/N0 {sys.argv[2]}: FETCH_SIZE per launch of 131072 frames (raw counter, KB): median {m:.0f} over {len(v)} launches = {m * 1024 * 2 / 131072:.0f} B per frame after the x2 correction (algorithmic input 5120 B)")
else:
    print("   no decode_ms_bs_kernel dispatch found in", sys.argv[1])
