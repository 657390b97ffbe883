nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print(len(os.sched_getaffinity(0)))"
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python __graft_entry__.py smoke 2>&1 | tail -2
for v in 4 2; do python bench.py --no-cpu --variant $v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slp v$v', d['value'], d['roofline']['kernel_ms'])"; done
export LABRADOR_LDPC_HIP_LIB=$PWD/labrador_ldpc_amd/liblabrador_ldpc_hip_noslp.so
for v in 4 2; do python bench.py --no-cpu --variant $v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('noslp v$v', d['value'], d['roofline']['kernel_ms'])"; done
