#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t6f_base t6f_2c t6f_no t12f_base t12f_no t2f_base t2f_no tcf_base tcf_no t6f_base t6f_no > gpurun_out/r03_kb21.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb21.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-250
