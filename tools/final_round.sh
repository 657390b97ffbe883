#!/bin/bash
# Everything a round's evidence directory holds, in one gpurun call:  bash tools/final_round.sh r05_final
#   -> gpurun_out/<round>/ ; condense with  python tools/summarise_round.py <round>  (-> profiles/<round>, profiles/hbm_traffic.json)
N=${1:-r06_final}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$N; mkdir -p $O
cd $R
bash tools/prof_round.sh $N > $O/prof_round.log 2>&1                      # bench line, kernel trace of the same command, PMC passes per config
python3 tools/rates_all.py > $O/rates_all_codes.txt 2> $O/rates_all_codes.err    # every code x LLR type, encode, decode_bf
python3 tools/bs_ab.py 262144 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/bs_rates.txt   # the bit-sliced i8 kernels at large batches
bash tools/prof_frow.sh $N > $O/prof_frow.log 2>&1                        # encode / decode_bf / awgn under rocprofv3
python3 tests/soak/big_soak.py 4 > $O/big_soak_x4.txt 2>&1               # 9 codes x 5 LLR types x 5 operating points against the oracle
python3 tests/soak/bs_soak.py 8 > $O/bs_soak_x8.txt 2>&1                 # the bit-sliced kernels against the oracle
python3 tests/soak/enc_soak.py > $O/enc_soak.txt 2>&1
python3 tools/enc_bench.py > $O/encoder_rates.txt 2>&1
python3 tools/hp_sweep.py > $O/host_path_rates.txt 2>&1                  # PCIe-inclusive host-pointer entry points
python3 tools/host_path_devices.py >> $O/host_path_rates.txt 2>&1
bash tools/ber_regression.sh $N                                          # nine BER sweeps, byte-identical to profiles/r01_ber, r02_ber
python3 tools/record_expected_jobs.py --check > $O/expected_jobs_check.txt 2>&1   # bench.EXPECTED_JOBS reproduced on THIS box and build
build/ub/wide_rate > $O/wide_rate.txt 2>&1                               # f64 / i32 issue costs (tools/ubench/wide_rate.hip, built by tools/r06_build_tools.sh)
python3 tools/wide_types.py 32768 TM8192 TM2048 2>&1 | grep -v amdgpu.ids > $O/wide_types.txt
cc -std=c11 -O1 -pthread -Iinclude tests/c/threads_single_frame.c -Llabrador_ldpc_amd -llabrador_ldpc_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/labrador_ldpc_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/thr &&
  (for n in 1 4 16; do /tmp/thr $n 2000 2:0; done; for n in 1 16; do /tmp/thr $n 500 8:0; done; /tmp/thr 16 540) > $O/single_frame_threads.txt 2>&1
python3 tests/soak/single_frame_latency.py 2>&1 | grep -v amdgpu.ids > $O/single_frame_latency.txt   # per-call latency of the reference-shaped entries
cc -std=c11 -O2 -Iinclude tools/ubench/single_frame_latency.c -Llabrador_ldpc_amd -llabrador_ldpc_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/labrador_ldpc_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/sfl &&
  (/tmp/sfl; echo "-- LABRADOR_LDPC_HIP_NO_NOTIFY=1"; LABRADOR_LDPC_HIP_NO_NOTIFY=1 /tmp/sfl) > $O/single_frame_latency_c.txt 2>&1    # ... from C, with and without the completion ticket
python3 tests/soak/refill_fuzz.py 300 2>&1 | grep -v amdgpu.ids > $O/refill_fuzz.txt                  # the slot-refill kernels on random batch sizes / caps / streams
tail -2 $O/big_soak_x4.txt; tail -1 $O/bs_soak_x8.txt; tail -3 $O/rates_all_codes.txt; tail -2 $O/expected_jobs_check.txt
