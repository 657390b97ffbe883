#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh p2f p2i8 l4_base t5_f2_4 t5_f5_4 l2_base t5_f2_2 t5_f5_2 t6i8_f0 t6i8_f2 t6i8_f5 t2i8_f2 t2i8_f3 t2i8_f5 t2_f2 t2_f3 t2_f5 tc_f2 tc_f3 tc_f5 t12i8_f0 t12i8_f5 > gpurun_out/r03_kb11.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb11.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-260
