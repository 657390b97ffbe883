run() { bash tools/kb.sh "-DKCODE=8 -DKFRAMES=131072 -DKGRID=256 -DKPAIR=1 $1" 2>&1 | grep -v "^\[-" | sed "s/^/$1: /"; }
run ""
run "-DLDPC_PAIR_ODD_B64=1"
run "-DLDPC_PRIO_ROWS_PAIR={3,3,2,2,2,0}"
run "-DLDPC_PRIO_ROWS_PAIR={3,2,2,2,1,0}"
run "-DLDPC_PRIO_ROWS_PAIR={3,3,3,2,1,0}"
run "-DLDPC_PRIO_ROWS_PAIR={2,2,2,1,1,0}"
