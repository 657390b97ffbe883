#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t6f_base t6f_nocap2 t6f_nocap5 t15f_base t15f_nocap2 > gpurun_out/r03_kb14.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb14.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-250
