#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t12f_l t12f_lp t12f_np1 t15i_l t15i_l5 t15i_l10 t2i_l t2i_l3 t12i_l t12i_l9 t12f_l t12f_lp > gpurun_out/r03_kb28.txt 2>&1
grep -E "^==|cw/s|occupancy" gpurun_out/r03_kb28.txt | paste - - - | sed 's/| mean iters/it/; s/success.*hash/hash/; s/ x 256 CUs//' | cut -c1-250
