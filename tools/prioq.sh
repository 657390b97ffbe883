sw() { # code ipt grid ebn0 type ks...
  c=$1; ipt=$2; g=$3; eb=$4; T=$5; shift 5
  for k in "$@"; do bash tools/kb.sh "-DKT=$T -DKCODE=$c -DKIPT=$ipt -DKFRAMES=262144 -DKGRID=$g -DKEBN0=$eb -DLDPC_LOCAL_IN_VAR=$k" 2>&1 | grep -v "^\[-" | sed "s/^/K=$k: /"; done
}
sw 7 1 256 3.0 float 0 3 5 7 9 12
sw 5 1 8192 2.5 float 0 2 3 4 5 7
sw 5 1 8192 2.5 int8_t 0 2 4
sw 4 1 65536 3.0 float 0 3 5
sw 3 1 65536 4.0 float 0 4 7
