#!/bin/bash
# Round 6, item 2 of round 5's review (config 5, decode_ms_bs_split_kernel<6>): where are the 20.97 M LDS bank-conflict cycles per launch,
# and what does a wave wait for?  PMC passes on 131 072 TM5120 i8 frames at 2 dB with max_iters 0 / 1 / 2 / 25.
O=$GRAFT_REPO_ROOT/gpurun_out/r06_cfg5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $O/list_avail.txt 2>&1
for mi in 0 1 2 25; do
  B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-configs --steps 2 --warmup 1 --code TM5120 --dtype i8 --frames-per-gpu 131072 --ebn0 2.0 --maxiters $mi"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/mi$mi.pmcA -- $B > $O/mi$mi.pmcA.log 2>&1
done
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-configs --steps 2 --warmup 1 --code TM5120 --dtype i8 --frames-per-gpu 131072 --ebn0 2.0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/mi25.pmcB -- $B > $O/mi25.pmcB.log 2>&1
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_WAVES_EQ_64 SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $O/mi25.pmcC -- $B > $O/mi25.pmcC.log 2>&1
ls $O
