#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "whole or config1" 2>&1 | tail -5 ) > gpurun_out/r03_c13_tests.log
tools/kb_run.sh p6_base p6_odd p6_odd_liv8 p6_odd_liv12 p6_base p6_odd p6_i8_odd0 p2i8 p6_stamps > gpurun_out/r03_kb13.txt 2>&1
tail -3 gpurun_out/r03_c13_tests.log; grep -E "^==|cw/s|wave |quarter|stamps" gpurun_out/r03_kb13.txt | sed 's/| mean iters/it/; s/success.*hash/hash/' | cut -c1-230
