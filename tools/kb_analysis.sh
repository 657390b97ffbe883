#!/bin/bash
# Where does the TM8192 f32 pair kernel's time go?  Same-process timing of the shipped kernel and of diagnostic
# builds (tools/kbench.hip defines LDPC_KBENCH, the only place LDPC_DIAG_* / tuning overrides are allowed).
# Run on the GPU box:  bash tools/kb_analysis.sh > gpurun_out/kbench_analysis.txt
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-fast-math -ffp-contract=off -fno-slp-vectorize -Wno-unused-value -Ibuild/kb_src"
kb() {   # label, flags...
  label=$1; shift
  $HIPCC -DKPAIR=1 -DKGRID=256 "$@" -o /tmp/kb_$$ tools/kbench/kbench.hip build/kb_src/channel.hip 2>&1 | grep -E "error" -A3
  printf "%-58s " "$label"; /tmp/kb_$$ | tail -${KB_TAIL:-1}
}
echo "== shipped kernel, early termination (2 dB, 65536 frames, persistent grid of 256 workgroups) =="
kb "shipped" -DKFRAMES=65536
KB_TAIL=31 kb "phase stamps (s_memtime; perturbs the timing)" -DKFRAMES=65536 -DLDPC_DIAG_STAMPS
kb "odd rotations as aligned 64-bit reads (LDPC_PAIR_ODD_B64=1)" -DKFRAMES=65536 -DLDPC_PAIR_ODD_B64=1
echo "== leave-one-out, 25 iterations for every frame (results wrong by construction) =="
for v in "" NOSELFCORR NOMIN NOSIGN NOPAR NOBARRIER; do
  d=""; [ -n "$v" ] && d="-DLDPC_DIAG_$v"
  kb "fixed 25 iterations ${v:-(everything in)}" -DKFRAMES=32768 -DLDPC_DIAG_FIXED_ITERS $d
done
echo "== two against four waves per SIMD on the (t, t + M/k) kernels (equal instruction counts per index) =="
for ipt in 2 4; do
  $HIPCC -DKCODE=8 -DKIPT=$ipt -DKGRID=256 -DKFRAMES=65536 -o /tmp/kb_$$ tools/kbench/kbench.hip build/kb_src/channel.hip 2>&1 | grep -E "error" -A3
  printf "%-58s " "$ipt indices per thread"; /tmp/kb_$$ | tail -1
done
rm -f /tmp/kb_$$
