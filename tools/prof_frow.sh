#!/bin/bash
# rocprofv3 on the f-row kernels (review item 7): kernel trace + three PMC passes of tools/frow_workload.py, with the bit-sliced decode_bf
# and with the byte-per-variable one.   usage (via gpurun): bash tools/prof_frow.sh r04_final   -> gpurun_out/r04_final/frow_*
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_final}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/frow_trace $O/frow_trace_bytes $O/frow_pmc[1-3]
rocprofv3 --kernel-trace --stats --output-format csv -d $O/frow_trace -- python3 $R/tools/frow_workload.py > $O/frow_workload.json 2> $O/frow_trace.err
LABRADOR_LDPC_HIP_BF_BYTES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/frow_trace_bytes -- python3 $R/tools/frow_workload.py > $O/frow_workload_bytes.json 2> $O/frow_trace_bytes.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/frow_pmc1 -- python3 $R/tools/frow_workload.py > /dev/null 2> $O/frow_pmc1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/frow_pmc2 -- python3 $R/tools/frow_workload.py > /dev/null 2> $O/frow_pmc2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/frow_pmc3 -- python3 $R/tools/frow_workload.py > /dev/null 2> $O/frow_pmc3.err
find $O -name "*kernel_stats.csv" | head; tail -1 $O/frow_workload.json | cut -c1-300
