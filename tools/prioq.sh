run() { bash tools/kb.sh "-DKCODE=8 -DKFRAMES=131072 -DKGRID=256 -DKPAIR=1 $1" 2>&1 | grep -v "^\[-" | sed "s/^/$1: /"; }
run ""
run "-DLDPC_PAIR_ODD_B64=1"
run ""
run "-DLDPC_PAIR_ODD_B64=1"
