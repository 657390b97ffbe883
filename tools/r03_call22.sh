#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "single or three_flip or config1 or c_client or example or batch_smoke or host or ragged or empty" 2>&1 | tail -4
python tools/single_frame_latency.py > gpurun_out/single_frame_latency.txt 2>&1
LABRADOR_LDPC_HIP_NO_DIRECT=1 python tools/single_frame_latency.py >> gpurun_out/single_frame_latency.txt 2>&1
cat gpurun_out/single_frame_latency.txt
