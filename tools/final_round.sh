#!/bin/bash
# Everything a round's evidence directory holds, in one gpurun call:  bash tools/final_round.sh r03_final
N=${1:-r03_final}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$N; mkdir -p $O
cd $R
bash tools/prof_round.sh $N > $O/prof_round.log 2>&1
python3 tools/rates_all.py > $O/rates_all_codes.txt 2> $O/rates_all_codes.err
bash tools/kb_analysis.sh > $O/kbench_analysis.txt 2>&1
python3 tests/soak/big_soak.py > $O/big_soak.txt 2>&1
python3 tools/enc_bench.py > $O/encoder_rates.txt 2>&1
python3 tools/hp_sweep.py > $O/host_path_rates.txt 2>&1
python3 tools/host_path_devices.py >> $O/host_path_rates.txt 2>&1
S=0.6,0.8,1.0,1.2,1.4,1.6,1.8,2.0,2.5,3.0,3.5,4.0
python3 -m labrador_ldpc_amd.perftest --code TM8192 --noise ebn0 --snrs $S --max-bits 1e9 --max-errors 50000 > $O/ber_TM8192.ebn0.csv 2> $O/ber.err
python3 -m labrador_ldpc_amd.perftest --code TM2048 --noise ebn0 --snrs $S --max-bits 1e9 --max-errors 50000 > $O/ber_TM2048.ebn0.csv 2>> $O/ber.err
python3 -m labrador_ldpc_amd.perftest --code TC512 --noise ebn0 --snrs $S --max-bits 1e9 --max-errors 50000 > $O/ber_TC512.ebn0.csv 2>> $O/ber.err
cmp $O/ber_TM8192.ebn0.csv profiles/r01_ber/TM8192.ebn0.csv && cmp $O/ber_TM2048.ebn0.csv profiles/r01_ber/TM2048.ebn0.csv && cmp $O/ber_TC512.ebn0.csv profiles/r01_ber/TC512.ebn0.csv && echo "BER sweeps (TM8192, TM2048, TC512) byte-identical to profiles/r01_ber" > $O/ber_regression.txt || echo "BER sweeps DIFFER from profiles/r01_ber" > $O/ber_regression.txt
bash tools/ber_regression_r02.sh >> $O/ber_regression.txt 2>&1          # the other six codes against profiles/r02_ber
cat $O/ber_regression.txt; tail -2 $O/big_soak.txt; tail -3 $O/rates_all_codes.txt
