#!/bin/bash
# tools/ab.sh "<variants>" "<configs>" [rounds] -- A/B of prebuilt libddx.so variants (ab/<name>.so) on the GPU box:
# interleaved rounds (a b a b ...) of `bench.py --no-cpu-baseline --no-extras` per config; one compact line per run.
# config syntax: name[@distance]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
variants=$1; configs=$2; rounds=${3:-2}
cp diffdope_amd/libddx.so /tmp/libddx_keep.so
for r in $(seq 1 $rounds); do
  for v in $variants; do
    cp ab/$v.so diffdope_amd/libddx.so
    for c in $configs; do
      name=${c%@*}; dist=""; [[ "$c" == *@* ]] && dist="--distance ${c#*@}"
      python bench.py --no-cpu-baseline --no-extras --config $name $dist $AB_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms']
    print('$v', '$c', round(d['value']), {n[:7]: round(x*1e3,1) for n,x in k.items() if x>0}, 'rot %.1e'%d['final_pose']['rot_err_rad_best'])
except Exception as e: print('$v $c ERR', e)"
    done
  done
done
cp /tmp/libddx_keep.so diffdope_amd/libddx.so
