#!/bin/bash
# Re-run the six BER sweeps of profiles/r02_ber on the current build; they must come out byte-identical.
#   bash tools/ber_regression_r02.sh > gpurun_out/ber_regression_r02.txt
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/ber_r02; mkdir -p $O
ok=1
for f in $R/profiles/r02_ber/*.ebn0.csv; do
  c=$(basename $f .ebn0.csv)
  snrs=$(cut -d, -f2 $f | paste -sd, -)
  python3 -m labrador_ldpc_amd.perftest --code $c --noise ebn0 --snrs $snrs --max-bits 1e9 --max-errors 50000 > $O/$c.ebn0.csv 2>> $O/err.txt
  if cmp -s $O/$c.ebn0.csv $f; then echo "$c: byte-identical to profiles/r02_ber ($(wc -l < $f) points)"; else echo "$c: DIFFERS"; ok=0; fi
done
[ $ok = 1 ] && echo "all six sweeps byte-identical" || echo "REGRESSION"
