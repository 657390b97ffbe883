run() { bash tools/kb.sh "-DKCODE=8 -DKFRAMES=131072 -DKGRID=256 -DKPAIR=1 $1" 2>&1 | grep -v "^\[-" | sed "s/^/$1: /"; }
run ""
run "-DLDPC_PAIR_HEAD_Q={2,2,3,3}"
run "-DLDPC_PAIR_HEAD_Q={2,3,3,3}"
run "-DLDPC_PAIR_VAR_Q={1,1,2,3} -DLDPC_PAIR_VAR2_Q={0,0,1,2}"
run "-DLDPC_PAIR_VAR_Q={1,2,2,3} -DLDPC_PAIR_VAR2_Q={0,0,0,1}"
run "-DLDPC_PAIR_VAR_Q={0,1,2,3} -DLDPC_PAIR_VAR2_Q={0,1,2,3}"
run "-DLDPC_PAIR_HEAD_Q={2,2,3,3} -DLDPC_PAIR_VAR_Q={1,1,2,3} -DLDPC_PAIR_VAR2_Q={0,0,1,2}"
