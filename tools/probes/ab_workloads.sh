#!/bin/bash
# A/B prebuilt library variants over several bench workloads: ab_workloads.sh "<tags>" "<bench arg sets separated by ;>"
IFS=';' read -ra SETS <<< "$2"
for v in $1; do
  cp tools/probes/libs/libspx_$v.so scaleprotoseg_amd/libspx_hip.so
  for s in "${SETS[@]}"; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $s 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernels']
        print('$v | $s | step %.3f ms | fwd %.3f  K1 %.3f  K2 %.3f' % (j['ms_per_step'], k['spx_dist_fwd']['ms'], k['spx_dist_bwd']['ms'], k['spx_bank_bwd']['ms']))
"
  done
done
