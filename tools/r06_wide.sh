#!/bin/bash
# Round 6, item 7: i32 / f64 on TM8192 -- issue-rate ceilings (tools/ubench/wide_rate.hip, built here: build/ub/wide_rate), rates, and the
# kernels' instruction counters.   gpurun -- bash tools/r06_wide.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r06_wide; mkdir -p $O; cd $GRAFT_REPO_ROOT
build/ub/wide_rate > $O/wide_rate.txt 2>&1
python3 tools/wide_types.py 32768 TM8192 TM2048 > $O/rates.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/tools/wide_types.py 32768 TM8192 > $O/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc -- python3 $GRAFT_REPO_ROOT/tools/wide_types.py 32768 TM8192 > $O/pmc.log 2>&1
cat $O/wide_rate.txt $O/rates.txt
