#!/bin/bash
# Run on the GPU box: the default bench once per library variant (A/B/A/B ...), one line each.
#   tools/ab_variants.sh [-r ROUNDS] [-a "bench args"] default nobar ...      ("default" = the product library)
rounds=2; args=""
while getopts "r:a:" o; do case $o in r) rounds=$OPTARG;; a) args=$OPTARG;; esac; done
shift $((OPTIND - 1))
for i in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset SPX_LIB_OVERRIDE; else export SPX_LIB_OVERRIDE=scaleprotoseg_amd/variants/libspx_$v.so; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-modes $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernels']
        print('%-14s step %.3f ms | fwd %.3f  K1 %.3f  K2+ %.3f' % ('$v', j['ms_per_step'], k['spx_dist_fwd']['ms'], k.get('spx_dist_bwd', {'ms': 0})['ms'], k.get('spx_bank_bwd', {'ms': 0})['ms']))
" || exit 1
  done
done
