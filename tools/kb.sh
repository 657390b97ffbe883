#!/bin/bash
# usage (via gpurun): bash tools/kb.sh "<extra -D flags>" ["<more flag sets>" ...]
for fl in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-fast-math -ffp-contract=off -fno-slp-vectorize -Wno-unused-value \
    -Ilabrador_ldpc_amd/csrc $fl -o /tmp/kbench tools/kbench.hip labrador_ldpc_amd/csrc/channel.hip 2>&1 | grep -E "error" -A3
  echo "[$fl]"; /tmp/kbench
done
