#!/bin/bash
# Profiling recipe of a round (run on the GPU box through gpurun): the default bench line, the same command under
# rocprofv3 --kernel-trace --stats, and PMC passes (separate runs, --pmc only) for the headline kernel and for the
# other BASELINE configs.   usage:  bash tools/prof_round.sh r03_final     -> gpurun_out/r03_final
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06_final}; mkdir -p $O
rm -rf $O/trace $O/*.pmc[1-5]          # a re-run must not leave an older run's files beside the new ones
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu > $O/trace_bench.log 2>&1
pmc() {   # tag, bench args...: five counter passes on short runs of one configuration
  tag=$1; shift
  B="python3 $R/bench.py --no-cpu --no-configs --steps 2 --warmup 1 $*"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/$tag.pmc1 -- $B > $O/$tag.pmc1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/$tag.pmc2 -- $B > $O/$tag.pmc2.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$tag.pmc3 -- $B > $O/$tag.pmc3.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$tag.pmc4 -- $B > $O/$tag.pmc4.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/$tag.pmc5 -- $B > $O/$tag.pmc5.log 2>&1
}
pmc TM8192_f32 --frames-per-gpu 65536
pmc TC512_f32  --code TC512  --frames-per-gpu 65536   --ebn0 2.0
pmc TM2048_f32 --code TM2048 --frames-per-gpu 262144  --ebn0 2.0
pmc TM5120_i8  --code TM5120 --dtype i8 --frames-per-gpu 131072 --ebn0 4.0
pmc TM5120_i8_2dB --code TM5120 --dtype i8 --frames-per-gpu 131072 --ebn0 2.0
pmc TM8192_i8 --code TM8192 --dtype i8 --frames-per-gpu 65536 --ebn0 2.0
find $O -name "*.csv" | wc -l
tail -1 $O/trace_bench.log | cut -c1-200
