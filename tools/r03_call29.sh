#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t12f_l t12f_np1 t12f_np1n t12f_ls t12f_np1s t12f_l t12f_np1n > gpurun_out/r03_kb29.txt 2>&1
grep -E "^==|cw/s|occupancy" gpurun_out/r03_kb29.txt | paste - - - | sed 's/| mean iters/it/; s/success.*hash/hash/; s/ x 256 CUs//' | cut -c1-250
