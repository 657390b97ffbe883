#!/bin/bash
mkdir -p gpurun_out
for F in 8192 32768 131072; do
  for n in t12f_ls t12f_l t12i_s t12i_l t15f_s t15f_b t15i_s t15i_l; do
    echo "== $n F=$F"; KB_F=$F KB_BURST=8 timeout 120 build/kb/$n
  done
done > gpurun_out/r03_kb31.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb31.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-200
