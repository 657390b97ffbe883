#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t12f_l t12f_c0 t5f_p1 t5f_p1_c0 t15f_b t15f_c0 t12f_l t12f_c0 > gpurun_out/r03_kb33.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb33.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-230
