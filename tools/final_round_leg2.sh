#!/bin/bash
# Second gpurun call of a round's evidence, AFTER tools/summarise_round.py has refreshed profiles/hbm_traffic.json for the build: the driver-shaped
# bench line (with counted traffic and VALU issue), the GPU suite, smoke, latency, the long soaks.  -> gpurun_out/r06_final2/
O=gpurun_out/r06_final2; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench.err
python3 -m pytest tests -m gpu -x -q > $O/gpu_tests_log.txt 2>&1; tail -3 $O/gpu_tests_log.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
cc -std=c11 -O2 -Iinclude tools/ubench/single_frame_latency.c -Llabrador_ldpc_amd -llabrador_ldpc_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/labrador_ldpc_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/sfl && (/tmp/sfl; echo "-- LABRADOR_LDPC_HIP_NO_NOTIFY=1"; LABRADOR_LDPC_HIP_NO_NOTIFY=1 /tmp/sfl) > $O/single_frame_latency_c.txt 2>&1
python3 tests/soak/single_frame_latency.py 2>&1 | grep -v amdgpu.ids > $O/single_frame_latency.txt
python3 tests/soak/refill_fuzz.py 300 2>&1 | grep -v amdgpu.ids > $O/refill_fuzz.txt; tail -1 $O/refill_fuzz.txt
python3 tests/soak/bs_soak.py 32 > $O/bs_soak_x32.txt 2>&1; tail -1 $O/bs_soak_x32.txt
python3 tests/soak/big_soak.py 12 > $O/big_soak_x12.txt 2>&1; tail -2 $O/big_soak_x12.txt
