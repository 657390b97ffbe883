#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh pair_base pair_1asm t2f_base t2f_1asm tcf_base tcf_1asm pair_base pair_1asm > gpurun_out/r03_kb26.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb26.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-260
