run() { bash tools/kb.sh "-DKT=int8_t -DKCODE=8 -DKFRAMES=131072 -DKGRID=256 -DKEBN0=3.0 -DKPAIR=1 $1" 2>&1 | grep -v "^\[-" | sed "s/^/$1: /"; }
for k in 0 4 7 10 14; do run "-DLDPC_PAIR_LOCAL_IN_VAR=$k"; done
