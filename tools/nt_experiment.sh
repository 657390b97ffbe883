#!/bin/bash
# Rate and FETCH_SIZE of the rate-4/5 bit-sliced kernel with the LLR planes from plane BS_NT_FROM on accessed non-temporally
# (tools/bs_diag_build.sh nt<k>:-DBS_NT_FROM=<k>).   usage (via gpurun): bash tools/nt_experiment.sh nt0 nt3 nt4 nt5 nt6
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f/nt; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
for v in product "$@"; do
  if [ $v = product ]; then unset LABRADOR_LDPC_HIP_LIB; else export LABRADOR_LDPC_HIP_LIB=$R/build/diag/liblabrador_ldpc_hip_$v.so; fi
  echo "== $v"
  python3 $R/tools/bs_diag.py 2>&1 | grep "TM5120\|TM1280"
  for e in 2.0 4.0; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$v.$e -- python3 $R/bench.py --no-cpu --no-configs --steps 2 --warmup 1 --code TM5120 --dtype i8 --frames-per-gpu 131072 --ebn0 $e > $O/$v.$e.log 2>&1
    python3 $R/tools/nt_fetch.py $O/$v.$e $e
  done
done
