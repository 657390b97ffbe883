#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t5f_p1 t5f_p1_no t5f_p1_no5 t5f_p1_2c t5f_p1 t5f_p1_no > gpurun_out/r03_kb23.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb23.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-260
