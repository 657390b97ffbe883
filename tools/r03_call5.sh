#!/bin/bash
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r03_c5_tests.log
tools/kb_run.sh x_t5f x_t5f_nl x_t12f x_tc > gpurun_out/r03_kb5.txt 2>&1
timeout 900 python tools/rates_all.py > gpurun_out/r03_rates_all_b.txt 2>&1
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/r03_bench_second.json 2> gpurun_out/r03_bench_second.err
tail -3 gpurun_out/r03_c5_tests.log; grep -E "^==|cw/s" gpurun_out/r03_kb5.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/'; cat gpurun_out/r03_rates_all_b.txt
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_second.json'))
print('headline', d['value'], d['ms_per_step'], d['parity'])
for k,v in d['configs'].items():
    print(k, {kk:vv for kk,vv in v.items() if kk in ('value','steps','ms_per_step','equal_to_cpu_oracle','parity')})
PY
