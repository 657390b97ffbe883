#!/bin/bash
# Cross-compile kbench variants HERE (no GPU needed) into build/kb/<name>; build/ travels with gpurun.
# usage: tools/kb_build.sh name "<-D flags>" [name "<flags>" ...]   (runs up to $KB_JOBS compiles in parallel, default 7)
cd "$(dirname "$0")/.." || exit 1
mkdir -p build/kb
jobs=${KB_JOBS:-7}
while [ $# -ge 2 ]; do
  name=$1; fl=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-fast-math -ffp-contract=off -fno-slp-vectorize -Wno-unused-value \
      -Ilabrador_ldpc_amd/csrc $fl -o build/kb/$name tools/kbench.hip labrador_ldpc_amd/csrc/channel.hip 2>&1 | grep -E "error" -A3
    echo "$fl" > build/kb/$name.flags ) &
  while [ "$(jobs -r | wc -l)" -ge "$jobs" ]; do sleep 0.5; done
done
wait
