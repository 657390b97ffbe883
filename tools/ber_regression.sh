#!/bin/bash
# All nine BER sweeps (TM8192 / TM2048 / TC512 of profiles/r01_ber, the other six of profiles/r02_ber) on the current build: they must
# come out byte-identical.   usage (via gpurun):  bash tools/ber_regression.sh r04_final   -> gpurun_out/r04_final/ber_regression.txt
N=${1:-r04_final}; R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/$N; mkdir -p $O
cd $R
S=0.6,0.8,1.0,1.2,1.4,1.6,1.8,2.0,2.5,3.0,3.5,4.0
ok=1
for c in TM8192 TM2048 TC512; do
  python3 -m labrador_ldpc_amd.perftest --code $c --noise ebn0 --snrs $S --max-bits 1e9 --max-errors 50000 > $O/ber_$c.ebn0.csv 2>> $O/ber.err
  cmp -s $O/ber_$c.ebn0.csv profiles/r01_ber/$c.ebn0.csv || ok=0
done
[ $ok = 1 ] && echo "BER sweeps (TM8192, TM2048, TC512) byte-identical to profiles/r01_ber" > $O/ber_regression.txt || echo "BER sweeps DIFFER from profiles/r01_ber" > $O/ber_regression.txt
bash tools/ber_regression_r02.sh >> $O/ber_regression.txt 2>&1
python3 -c "import labrador_ldpc_amd as l; print('library build', l.lib.labrador_ldpc_hip_build_id().decode())" >> $O/ber_regression.txt 2>/dev/null
cat $O/ber_regression.txt
