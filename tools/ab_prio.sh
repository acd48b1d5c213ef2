# same-box A/B of T3D_X3_PRIO_WGRAD (weight-gradient tiles at wave priority 1): variant tools/build_x3_variant.sh noprio "-DT3D_X3_PRIO_WGRAD=0"
echo "== timeline of the fused backward, default build"
timeout 600 python tools/trace_bwd.py 2>&1 | grep -v amdgpu.ids
for v in "" tools/libt3d_noprio.so; do
  echo "== launches ${v:-default (wgrad tiles at priority 1)}"; T3D_LIB=$v timeout 300 python tools/bench_x3.py 2>&1 | grep "^bwd" | sed 's/fp32-MFMA.*x3:/x3:/'
done
for i in 1 2 3 4; do for v in "" tools/libt3d_noprio.so; do
T3D_LIB=$v python bench.py --steps 200 --warmup 30 --no_other_configs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('step', '${v:-default}', d['ms_per_step'])"
done; done
