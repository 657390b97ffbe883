#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t2f_base t2f_pair t2f_pair_s t2f_base t2f_pair > gpurun_out/r03_kb24.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb24.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-260
