#!/bin/bash
# Everything a round's evidence directory holds, in one gpurun call:  bash tools/final_round.sh r05_final
#   -> gpurun_out/<round>/ ; condense with  python tools/summarise_round.py <round>  (-> profiles/<round>, profiles/hbm_traffic.json)
N=${1:-r05_final}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$N; mkdir -p $O
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
tail -2 $O/big_soak_x4.txt; tail -1 $O/bs_soak_x8.txt; tail -3 $O/rates_all_codes.txt
