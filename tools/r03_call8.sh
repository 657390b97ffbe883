#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh l4_base lv4 l2_base lv2 > gpurun_out/r03_kb8.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb8.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/'
