sw() { c=$1; ipt=$2; g=$3; eb=$4; T=$5; shift 5
  for k in "$@"; do bash tools/kb.sh "-DKT=$T -DKCODE=$c -DKIPT=$ipt -DKFRAMES=262144 -DKGRID=$g -DKEBN0=$eb -DLDPC_LOCAL_IN_VAR=$k" 2>&1 | grep -v "^\[-" | sed "s/^/K=$k: /"; done; }
sw 7 1 256 3.0 int8_t 0 2 4 6 9
