# scratch: A/B prebuilt libspx_hip.so variants (tools/probes/libs/libspx_<tag>.so) on the group-phase workload (4-scale bank, fused tail)
for v in "$@"; do
  cp tools/probes/libs/libspx_$v.so scaleprotoseg_amd/libspx_hip.so
  echo "== $v"
  timeout -k 10 200 python bench.py --workload cityscapes_1024x2048_c256_p228_s4 --group-tail --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernels']
        print('step %.3f ms | fwd %.3f  K1 %.3f  K2 %.3f' % (j['ms_per_step'], k['spx_dist_fwd']['ms'], k['spx_dist_bwd']['ms'], k['spx_bank_bwd']['ms']))
"
done
