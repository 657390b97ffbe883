#!/bin/bash
mkdir -p gpurun_out
build/kb/valu_rate > gpurun_out/r03_valu_rate.txt 2>&1
tools/kb_run.sh p3_base p4_fma > gpurun_out/r03_kb9.txt 2>&1
grep -E "fmac|fma_f32|med3|legacy|edge update|v_mul_f32|v_cmp_lt|v_cndmask|v_min3|v_xor|bitop3" gpurun_out/r03_valu_rate.txt; grep -E "^==|cw/s" gpurun_out/r03_kb9.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/'
