#!/bin/bash
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_goldens.py -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r03_c4_tests.log
tools/kb_run.sh x_t5f x_t5f_nonan x_t12f x_t12f_nonan x_pair x_pair_nonan x_t2f x_t2f_nonan x_tc x_tc_nonan x_tc_nocap x_tc_nocap_m3 x_t6f x_t6f_nonan \
   t5_4_now t5_2_now t12i8_now t12i8_nopk > gpurun_out/r03_kb4.txt 2>&1
{ for v in x_tc x_tc_nocap x_tc_nocap_m3; do echo "== $v F=65536 burst=8"; KB_F=65536 KB_BURST=8 build/kb/$v | grep cw/s; done; } >> gpurun_out/r03_kb4.txt 2>&1
timeout 900 python tools/rates_all.py > gpurun_out/r03_rates_all_a.txt 2>&1
tail -3 gpurun_out/r03_c4_tests.log; grep -E "^==|cw/s" gpurun_out/r03_kb4.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/'; cat gpurun_out/r03_rates_all_a.txt
