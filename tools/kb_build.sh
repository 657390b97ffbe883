#!/bin/bash
# Cross-compile kbench variants HERE (no GPU needed) into build/kb/<name>; build/ travels with gpurun.
# usage: tools/kb_build.sh name "<-D flags>" [name "<flags>" ...]   (runs up to $KB_JOBS compiles in parallel, default 7)
#
# The kernels' experiment switches and timing diagnostics (LDPC_DIAG_*: leave a piece of the decoder out; LDPC_PAIR_ARRIVE_WAIT,
# LDPC_PAIR_LLR_DMA, LDPC_PAIR_FLAG_SETS, ...: measured and dropped alternatives) are NOT in the library's sources: they are the overlay
# tools/kbench/diag_overlay.patch, applied here to a COPY of labrador_ldpc_amd/csrc (build/kb_src), against which kbench is compiled.
cd "$(dirname "$0")/.." || exit 1
mkdir -p build/kb
rm -rf build/kb_src && cp -r labrador_ldpc_amd/csrc build/kb_src && rm -rf build/kb_src/profiles
if ! patch -s -p1 -d build/kb_src < tools/kbench/diag_overlay.patch; then
  echo "tools/kbench/diag_overlay.patch no longer applies to labrador_ldpc_amd/csrc (the f32-pipe kernels were edited): refresh it" >&2
  exit 1
fi
jobs=${KB_JOBS:-7}
while [ $# -ge 2 ]; do
  name=$1; fl=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-fast-math -ffp-contract=off -fno-slp-vectorize -Wno-unused-value \
      -Ibuild/kb_src $fl -o build/kb/$name tools/kbench/kbench.hip build/kb_src/channel.hip 2>&1 | grep -E "error" -A3
    echo "$fl" > build/kb/$name.flags ) &
  while [ "$(jobs -r | wc -l)" -ge "$jobs" ]; do sleep 0.5; done
done
wait
