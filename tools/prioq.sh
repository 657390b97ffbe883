run() { bash tools/kb.sh "-DKCODE=8 -DKFRAMES=131072 -DKGRID=256 -DKPAIR=1 $1" 2>&1 | grep -v "^\[-" | sed "s/^/$1: /"; }
run ""
run "-DLDPC_PAIR_QMAP={2,0,3,1}"
run "-DLDPC_PAIR_QMAP={2,3,0,1}"
run "-DLDPC_PAIR_QMAP={1,3,0,2}"
run "-DLDPC_PAIR_QMAP={3,2,1,0}"
run "-DLDPC_PAIR_QMAP={2,0,3,1} -DLDPC_DIAG_STAMPS"
