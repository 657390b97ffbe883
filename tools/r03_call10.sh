#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh p3_base p4_fma p5 p5_liv8 p5_liv12 pi8_liv8 p4_i8 p5_i8 l4_base t5_f5_4 l2_base t5_f5_2 t2_f3 t2_f5 x_tc tc_f5 > gpurun_out/r03_kb10.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb10.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/'
