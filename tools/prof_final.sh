#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + PMC passes -> gpurun_out/r01_final
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu > $O/trace_bench.log 2>&1
B="python3 $R/bench.py --no-cpu --frames-per-gpu 65536 --steps 2 --warmup 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc1 -- $B > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/pmc2 -- $B > $O/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -- $B > $O/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc4 -- $B > $O/pmc4.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc5 -- $B > $O/pmc5.log 2>&1
find $O -name "*.csv" | head -40
tail -1 $O/trace_bench.log | cut -c1-200
