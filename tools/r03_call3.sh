#!/bin/bash
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r03_c3_tests.log
tools/kb_run.sh t5_4_base t5_4_lv t5_4_pk t5_4_lvpk t5_2_base t5_2_lv t5_2_pk t5f_4_base t5f_4_lv > gpurun_out/r03_kb3.txt 2>&1
{
for g in 0 65536 32768; do for b in 1 8; do echo "== tcs_k1 F=65536 grid=$g burst=$b"; KB_F=65536 KB_GRID=$g KB_BURST=$b build/kb/tcs_k1 | grep "cw/s"; done; done
} > gpurun_out/r03_tc_grid.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for b in 1 8; do
  KB_F=65536 KB_BURST=$b rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_tc_trace_b$b -- $GRAFT_REPO_ROOT/build/kb/tcs_k1 > $GRAFT_REPO_ROOT/gpurun_out/r03_tc_trace_b$b.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for b in (1, 8):
    fs = glob.glob(f"gpurun_out/r03_tc_trace_b{b}/**/*kernel_trace.csv", recursive=True)
    if not fs: print("no trace", b); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "decode_ms_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    prev = None
    out = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.append((e - s, (s - prev) if prev else 0))
        prev = e
    print(f"burst {b}: (duration ns, gap to previous kernel ns):", out)
PY
tail -4 gpurun_out/r03_c3_tests.log; grep -E "^==|cw/s" gpurun_out/r03_kb3.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/'; cat gpurun_out/r03_tc_grid.txt | paste - - | sed 's/code 2 T4 ipt 1 pf 0//; s/| mean.*//'
