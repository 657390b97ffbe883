#!/bin/bash
# Cross-compile (no GPU needed) the small tools the round's evidence run uses; build/ travels with gpurun.
cd "$(dirname "$0")/.." && mkdir -p build/ub && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o build/ub/wide_rate tools/ubench/wide_rate.hip
