#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t5i8_b t5i8_x t5i8_f6 t5i8_b2 t5i8_x2 t5i8_f6_2 p8i8_f2 p8i8_f6 t2i8_f2 t2i8_f6 t12i8_f5 t12i8_f6 t15i8_f2 t15i8_f6 t6i8_f2 t6i8_f6 tc5i8_f2 tc5i8_f6 tc1i8_f2 tc1i8_f6 tc2i8_f2 tc2i8_f6 t5i8_b t5i8_x t5i8_f6 p8i8_f2 p8i8_f6 > gpurun_out/r03_kb18.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb18.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/; s/-DKIPT=1 -DKT=int8_t//' | cut -c1-230
