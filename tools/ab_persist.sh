# same-box A/B of T3D_W8_PERSIST (eight-wave forward kernels walk their tiles themselves): variant tools/build_x3_variant.sh nopersist "-DT3D_W8_PERSIST=0"
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "x3 or wide or eight or pool" 2>&1 | tail -2
for v in "" tools/libt3d_nopersist.so; do
  echo "== launches ${v:-default (persistent)}"; T3D_LIB=$v timeout 300 python tools/bench_x3.py 2>&1 | grep "^fwd" | sed 's/fp32-MFMA.*x3:/x3:/'
done
for i in 1 2 3; do for v in "" tools/libt3d_nopersist.so; do
T3D_LIB=$v python bench.py --steps 200 --warmup 30 --no_other_configs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('step', '${v:-default}', d['ms_per_step'])"
done; done
