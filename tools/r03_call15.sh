#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nan_two_pass or corner or small_maxiters or queue_fed" 2>&1 | tail -5
timeout 600 python tools/nan_two_pass_ab.py > gpurun_out/nan_two_pass_ab.txt 2>&1
cat gpurun_out/nan_two_pass_ab.txt
