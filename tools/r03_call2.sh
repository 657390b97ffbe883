#!/bin/bash
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_goldens.py -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r03_c2_tests.log
{
for v in tcs_k1 tcs_k8 tcd_k2 tcd_k4 tcd_k8 tcs_k1_fix tcd_k4_fix; do
  for f in 3072 12288 65536 262144 1048576; do
    for b in 1 8; do echo "== $v F=$f burst=$b"; KB_F=$f KB_BURST=$b timeout 120 build/kb/$v | grep "cw/s"; done
  done
done
} > gpurun_out/r03_tc_sweep.txt 2>&1
tools/kb_run.sh p_dyn_m2 p2_liv6 p2_liv7 p2_liv9 p2_liv10 p2_liv12 p2_oddb64 pi8_dyn_m2 pi8_liv2 pi8_liv6 pi8_liv8 \
   t2i8_0 t2i8_m2 t2i8_m3 t6i8_0 t6i8_m2 t15i8_0 t15i8_m2 tci8_0 tci8_m2 t12i8_0 t12i8_m2 > gpurun_out/r03_kb2.txt 2>&1
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/r03_bench_first.json 2> gpurun_out/r03_bench_first.err
tail -4 gpurun_out/r03_c2_tests.log; cat gpurun_out/r03_tc_sweep.txt | paste - - | sed 's/code 2 T4 ipt 1 pf 0//; s/| mean.*//' ; grep -E "^==|cw/s|QUEUE" gpurun_out/r03_kb2.txt | paste - - | sed 's/| mean.*hash/hash/'; cut -c1-1500 gpurun_out/r03_bench_first.json; tail -3 gpurun_out/r03_bench_first.err
