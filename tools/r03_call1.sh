#!/bin/bash
# GPU call 1 of round 3: parity of the new kernels, then kbench A/B, LDS widths, whole config 4
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_goldens.py -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r03_c1_tests.log
build/kb/lds_rate > gpurun_out/r03_lds_rate.txt 2>&1
tools/kb_run.sh p_static p_dyn p_dyn_m2 p_dyn_m3 pi8_static pi8_dyn pi8_dyn_m2 pi8_dyn_m3 \
   tc_static tc_dyn tc_dyn_k1 tc_dyn_k4 tc_dyn_k16 tc_big_static tc_big_dyn \
   t2_static t2_dyn t2_dyn_m2 t2_dyn_m3 t5_4_static t5_4_dyn t5_4_m2 t5_4_m3 t5_2_static t5_2_dyn t5_2_m2 t5_2_m3 > gpurun_out/r03_kb1.txt 2>&1
timeout 600 python tools/r03_fullbatch.py > gpurun_out/r03_fullbatch.log 2>&1
tail -5 gpurun_out/r03_c1_tests.log; cat gpurun_out/r03_lds_rate.txt; grep -E "^==|cw/s|QUEUE" gpurun_out/r03_kb1.txt; tail -4 gpurun_out/r03_fullbatch.log
