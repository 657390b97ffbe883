#!/bin/bash
# Alternative builds of the bit-sliced i8 kernels beside the product objects, for same-box A/B runs (tools/bs_ab.py):
#   bash tools/bs_alt_build.sh <name>:"<-D switches of decode_ms_bitslice.hpp / decode_ms_bs.hip>" ...
#   e.g.  w2:-DBS_WAVES_R12=2   p6:"-DBS_PINNED_R12=6"
# Both units of decode_ms_bs.hip are rebuilt (unit 2 with the product's scheduler flag unless BS2_FLAGS is set in the environment).
# -> build/alt/liblabrador_ldpc_hip_<name>.so; on the GPU:  LABRADOR_LDPC_HIP_LIB=$PWD/build/alt/liblabrador_ldpc_hip_<name>.so python tools/bs_ab.py
# The library's build id is that of the PRODUCT sources (capi.o is shared): bench.py refuses to stamp profile figures when
# LABRADOR_LDPC_HIP_LIB points elsewhere than the in-tree library.
set -e
R=$(cd $(dirname $0)/.. && pwd); O=$R/build/csrc; D=$R/build/alt; mkdir -p $D
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-fast-math -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
for spec in "$@"; do
  name=${spec%%:*}; extra=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $BS1_FLAGS $extra -DBS_TU=1 -I$O -c $R/labrador_ldpc_amd/csrc/decode_ms_bs.hip -o $D/decode_ms_bs_1_$name.o &&
    /opt/rocm/bin/hipcc $FLAGS ${BS2_FLAGS--mllvm -amdgpu-sched-strategy=iterative-ilp} $extra -DBS_TU=2 -I$O -c $R/labrador_ldpc_amd/csrc/decode_ms_bs.hip -o $D/decode_ms_bs_2_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -no-hip-rt -o $D/liblabrador_ldpc_hip_$name.so $(ls $O/*.o | grep -v "decode_ms_bs_[12].o") $D/decode_ms_bs_1_$name.o $D/decode_ms_bs_2_$name.o ) &
done
wait; ls -la $D/*.so
