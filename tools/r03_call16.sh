#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh pair_base pair_nocaponly pair_base pair_nocaponly > gpurun_out/r03_kb17.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb17.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nan_two_pass or corner" 2>&1 | tail -3
timeout 600 python tools/nan_two_pass_ab.py > gpurun_out/nan_two_pass_ab.txt 2>&1
cat gpurun_out/nan_two_pass_ab.txt
