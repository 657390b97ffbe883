#!/bin/bash
# On the GPU box: run precompiled kbench binaries (tools/kb_build.sh) in the given order; usage: tools/kb_run.sh name [name ...]
for n in "$@"; do
  echo "== $n [$(cat build/kb/$n.flags 2>/dev/null)]"
  timeout 300 build/kb/$n
done
