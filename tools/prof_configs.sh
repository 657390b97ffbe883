#!/bin/bash
# BASELINE.json configs 1, 2 and 4 (TC512 f32 65 536 frames, TM2048 f32 1 048 576 frames, TM5120 i8) under
# rocprofv3: kernel trace + HBM traffic counters in separate passes -> gpurun_out/r01_configs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01_configs; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args
  name=$1; shift
  python3 $R/bench.py --no-cpu "$@" > $O/$name.bench.json 2> $O/$name.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name.trace -- python3 $R/bench.py --no-cpu "$@" > $O/$name.trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$name.fetch -- python3 $R/bench.py --no-cpu --steps 2 "$@" > $O/$name.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$name.write -- python3 $R/bench.py --no-cpu --steps 2 "$@" > $O/$name.write.log 2>&1
}
run tc512_f32  --code TC512  --frames-per-gpu 65536   --ebn0 3.0
run tm2048_f32 --code TM2048 --frames-per-gpu 1048576 --ebn0 2.0
run tm5120_i8  --code TM5120 --dtype i8 --frames-per-gpu 524288 --ebn0 4.0
find $O -name "*.csv" | wc -l
