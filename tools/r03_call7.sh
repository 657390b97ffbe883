#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh l4_base l4_ch9 l4_ch3 l4_pA l4_pB l4_pC l4_p0 l2_base l2_ch9 l2_pA l2_pC l2_p0 p3_base p3_rA p3_rB p3_rC p3_rD p3_rE p3_v1 p3_v3 > gpurun_out/r03_kb7.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb7.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/'
