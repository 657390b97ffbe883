#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t2f_base t2f_pair t2f_pair4 t2f_base t2f_pair4 > gpurun_out/r03_kb25.txt 2>&1
grep -E "^==|cw/s|occupancy" gpurun_out/r03_kb25.txt | paste - - - | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-300
