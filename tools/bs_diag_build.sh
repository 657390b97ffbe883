#!/bin/bash
# Experimental builds of the bit-sliced i8 kernel beside the product objects: build/diag/liblabrador_ldpc_hip_<name>.so
#   <N>            BS_DIAG=<N>: timing-only (WRONG results): bit 1 = no LLR-plane global loads (rate 4/5), 2 = no permutation-table
#                  reads, 4 = no ds_bpermute
#   <name>:<flags> any other -D switches of decode_ms_bitslice.hpp, e.g.  nt5:-DBS_NT_FROM=5  (correct results)
#   bash tools/bs_diag_build.sh 1 2 4 7 nt5:-DBS_NT_FROM=5
# then on the GPU:  LABRADOR_LDPC_HIP_LIB=build/diag/liblabrador_ldpc_hip_diag1.so python tools/bs_diag.py
set -e
R=$(cd $(dirname $0)/.. && pwd); O=$R/build/csrc; D=$R/build/diag; mkdir -p $D
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-fast-math -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
for spec in "$@"; do
  case "$spec" in
    *:*) name=${spec%%:*}; extra=${spec#*:} ;;
    *)   name=diag$spec; extra="-DBS_DIAG=$spec -DBS_DIAG_BUILD" ;;
  esac
  ( /opt/rocm/bin/hipcc $FLAGS $extra -I$O -c $R/labrador_ldpc_amd/csrc/decode_ms_bs.hip -o $D/decode_ms_bs_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -no-hip-rt -o $D/liblabrador_ldpc_hip_$name.so $(ls $O/*.o | grep -v decode_ms_bs.o) $D/decode_ms_bs_$name.o ) &
done
wait; ls -la $D/*.so
