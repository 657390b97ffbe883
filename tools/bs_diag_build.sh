#!/bin/bash
# Diagnostic builds of the bit-sliced i8 kernel (wrong results, timing only): BS_DIAG bit 1 = no LLR-plane global loads (rate 4/5),
# 2 = no permutation-table reads, 4 = no ds_bpermute.  Writes build/diag/liblabrador_ldpc_hip_diag<N>.so beside the product objects.
#   bash tools/bs_diag_build.sh 1 2 4 7     then on the GPU:  LABRADOR_LDPC_HIP_LIB=build/diag/liblabrador_ldpc_hip_diag1.so python tools/bs_diag.py
set -e
R=$(cd $(dirname $0)/.. && pwd); O=$R/build/csrc; D=$R/build/diag; mkdir -p $D
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-fast-math -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DBS_DIAG=$n -DBS_DIAG_BUILD -I$O -c $R/labrador_ldpc_amd/csrc/decode_ms_bs.hip -o $D/decode_ms_bs_diag$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -no-hip-rt -o $D/liblabrador_ldpc_hip_diag$n.so $(ls $O/*.o | grep -v decode_ms_bs.o) $D/decode_ms_bs_diag$n.o ) &
done
wait; ls -la $D/*.so
