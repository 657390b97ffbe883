#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t5i8_f6 t5i8_sm t5i8_f6_2 t5i8_sm_2 p8i8_f6 p8i8_sm t2i8_f6 t2i8_sm t5i8_f6 t5i8_sm > gpurun_out/r03_kb19.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb19.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/; s/-DKIPT=1 -DKT=int8_t//' | cut -c1-230
