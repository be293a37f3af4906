# development A/B on one box: stamped variants (phase clocks), then the production variant t against the two-kernel backward
for v in $VARIANTS; do echo "== $v"; SPX_LIB_OVERRIDE=scaleprotoseg_amd/variants/libspx_$v.so timeout -k 10 200 python tools/diag_stamps_fused.py 2>&1 | grep -E "fused backward|wave 0|wave 4"; done
cp scaleprotoseg_amd/variants/libspx_t.so scaleprotoseg_amd/libspx_hip.so
timeout -k 10 200 python -m pytest tests/test_gpu_fused_bwd.py -q -k "x_dtype1-shape0 or x_dtype1-shape6 or run_to_run" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused   ', d['ms_per_step'], d['kernels'])"
SPX_FUSED_BWD=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2-kernel', d['ms_per_step'], d['kernels'])"
