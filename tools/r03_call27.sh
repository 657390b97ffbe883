#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t12f_base t12f_l t12i_b t12i_l t15f_b t15f_l t15i_b t15i_l t2f_base t2f_l t2i_b t2i_l t6f_base t6f_l t6i_b t6i_l > gpurun_out/r03_kb27.txt 2>&1
grep -E "^==|cw/s|occupancy" gpurun_out/r03_kb27.txt | paste - - - | sed 's/| mean iters/it/; s/success.*hash/hash/; s/ x 256 CUs//' | cut -c1-250
