#!/bin/bash
# Round 6, item 3: the pair kernel's odd-rotation STORES as one ds_write_b64 (LDPC_PAIR_ODD_ST64) against today's two ds_write_b32.
# Built here by:  tools/kb_build.sh st0_f32 "-DKPAIR=1 -DLDPC_PAIR_ODD_ST64=0" st1_f32 "-DKPAIR=1 -DLDPC_PAIR_ODD_ST64=1" (+ -DKT=int8_t for i8)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_st64; mkdir -p $O; cd $GRAFT_REPO_ROOT
bash tools/kb_run.sh st0_f32 st1_f32 st0_f32 st1_f32 st0_f32 st1_f32 st0_i8 st1_i8 st0_i8 st1_i8 > $O/ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for n in st0_f32 st1_f32 st0_i8 st1_i8; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/$n.pmc -- $GRAFT_REPO_ROOT/build/kb/$n > $O/$n.pmc.log 2>&1
done
grep -h "M cw/s" $O/ab.txt
