#!/bin/sh
# Identity of a library build: a hash of everything that determines the code object -- every source of this directory
# (*.hip *.hpp *.cpp, by name and content), the public header, the compiler flags and the compiler's version --
# NOT of the produced bytes: hipcc's output is not byte-deterministic (two builds of one tree differ in ~100 k bytes),
# so a hash of the .so calls a rebuild of identical sources "another build" (round 3's review, weak #5).
#   usage: build_id.sh <csrc dir> <public header> "<arch + compiler flags>" <hipcc>      -> 16 hex digits
# The Makefile bakes the result into the library (labrador_ldpc_hip_build_id()); bench.py ties a committed profile to it.
set -e
dir=$1; hdr=$2; flags=$3; hipcc=${4:-/opt/rocm/bin/hipcc}
{
  for f in $(cd "$dir" && ls *.hip *.hpp *.cpp 2>/dev/null | LC_ALL=C sort); do
    printf 'file %s\n' "$f"; cat "$dir/$f"
  done
  printf 'file %s\n' "$(basename "$hdr")"; cat "$hdr"
  printf 'flags %s\n' "$flags"
  "$hipcc" --version 2>/dev/null | grep -E 'HIP version|clang version|Target' || true
} | sha256sum | cut -c1-16
