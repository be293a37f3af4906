set -e
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-modes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-modes --outputs logits 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 logits-only', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"; }
cp scaleprotoseg_amd/libspx_hip.so /tmp/base.so
run base
cp scaleprotoseg_amd/variants/libspx_trow32.so scaleprotoseg_amd/libspx_hip.so
run trow32
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward or push or large or random_conf" 2>&1 | tail -1
cp /tmp/base.so scaleprotoseg_amd/libspx_hip.so
run base
cp scaleprotoseg_amd/variants/libspx_trow32.so scaleprotoseg_amd/libspx_hip.so
run trow32
