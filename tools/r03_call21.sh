#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh p8i8_old p8i8_pk p8i8_pk0 p8i8_old p8i8_pk p8i8_pk0 > gpurun_out/r03_kb22.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb22.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-250
