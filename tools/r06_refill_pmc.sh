#!/bin/bash
# Counters of the slot-refill kernels against their lockstep forms (TM1280 / TM1536 i8 at large batches): separate --pmc passes,
# bench.py's one-configuration mode.   -> gpurun_out/refill_pmc/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refill_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "TM1280 4.0 1048576 0" "TM1280 4.0 1048576 320" "TM1536 3.0 1048576 0" "TM1536 3.0 1048576 320"; do
  set -- $spec; tag=$1_v$4
  B="python3 $R/bench.py --no-cpu --no-configs --steps 2 --warmup 1 --code $1 --dtype i8 --ebn0 $2 --frames-per-gpu $3 --variant $4"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag.trace -- $B > $O/$tag.trace.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/$tag.pmc1 -- $B > $O/$tag.pmc1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/$tag.pmc2 -- $B > $O/$tag.pmc2.log 2>&1
  tail -1 $O/$tag.trace.log | cut -c1-160
done
find $O -name "*.csv" | wc -l
