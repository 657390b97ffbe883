#!/bin/bash
mkdir -p gpurun_out
( timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r03_c6_tests.log
make -C harness > /dev/null 2>&1
( timeout 900 python tools/host_path_devices.py 2>&1 | tail -30 ) > gpurun_out/r03_host_path.txt
tail -5 gpurun_out/r03_c6_tests.log; cat gpurun_out/r03_host_path.txt; nproc; cat /sys/bus/pci/devices/*/local_cpulist 2>/dev/null | sort | uniq -c | head -5
