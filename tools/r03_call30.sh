#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t12f_ls t12f_l t12f_k8 t12f_k16 t12i_s t12i_l t12i_k8 t12i_k16 t15f_s t15f_b t15f_k8 t15i_s t15i_l t15i_k8 > gpurun_out/r03_kb30.txt 2>&1
grep -E "^==|cw/s|occupancy" gpurun_out/r03_kb30.txt | paste - - - | sed 's/| mean iters/it/; s/success.*hash/hash/; s/ x 256 CUs//; s/-DKIPT=1 //; s/-DKFRAMES=1048576 //' | cut -c1-230
