#!/bin/bash
# Same-box A/B of the completion-ticket build against the library of the previous commit (build/alt/liblabrador_ldpc_hip_head.so, built from
# `git archive HEAD~` with OBJDIR / LIB overrides): the headline bench (the kernels gained two arguments), single-frame latency, thread client.
mkdir -p gpurun_out/sf
for i in 1 2 3; do
  python3 bench.py --no-cpu --no-configs --steps 10 --warmup 3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['ms_per_step'])"
  LABRADOR_LDPC_HIP_LIB=$PWD/build/alt/liblabrador_ldpc_hip_head.so python3 bench.py --no-cpu --no-configs --steps 10 --warmup 3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('head', d['value'], d['ms_per_step'])"
done 2>&1 | grep -v amdgpu | tee gpurun_out/sf/headline_ab.txt
for l in new head new head; do if [ $l = head ]; then export LABRADOR_LDPC_HIP_LIB=$PWD/build/alt/liblabrador_ldpc_hip_head.so; else unset LABRADOR_LDPC_HIP_LIB; fi
  python3 bench.py --no-cpu --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', {k: round(v['value'] / 1e6, 2) for k, v in d['configs'].items() if v.get('value')})"; done | tee gpurun_out/sf/configs_ab.txt; unset LABRADOR_LDPC_HIP_LIB
python3 tests/soak/single_frame_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sf/latency_notify2.txt
LABRADOR_LDPC_HIP_NO_NOTIFY=1 python3 tests/soak/single_frame_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sf/latency_no_notify2.txt
for l in new head new head; do if [ $l = head ]; then export LABRADOR_LDPC_HIP_LIB=$PWD/build/alt/liblabrador_ldpc_hip_head.so; else unset LABRADOR_LDPC_HIP_LIB; fi; echo "== $l"; python3 tools/rates_all.py 2>/dev/null | grep -E "^TC|^TM1|^TM2048|code"; done | tee gpurun_out/sf/rates_ab.txt; unset LABRADOR_LDPC_HIP_LIB
CC="cc -std=c11 -O1 -pthread -Iinclude tests/c/threads_single_frame.c"
$CC -Llabrador_ldpc_amd -llabrador_ldpc_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/labrador_ldpc_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/thr_new
$CC -Lbuild/alt -l:liblabrador_ldpc_hip_head.so -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/build/alt -Wl,-rpath,/opt/rocm/lib -o /tmp/thr_head
(for r in 1 2; do for b in new head; do echo "== $b"; /tmp/thr_$b 16 540; /tmp/thr_$b 16 2000 2:0; /tmp/thr_$b 1 2000 2:0; /tmp/thr_$b 16 500 8:0; done; done) 2>&1 | tee gpurun_out/sf/threads_ab.txt
