# same-box A/B of the accumulator placement of the x3 kernels (T3D_X3_AGPR): per-shape launch times, then the whole step
export T3D_PC_MODES=0
echo "== AGPR accumulators (default build)"; timeout 300 python tools/bench_x3_pc.py 2>&1 | grep -v amdgpu.ids
echo "== VGPR accumulators (-DT3D_X3_AGPR=0)"; T3D_LIB=tools/libt3d_noagpr.so timeout 300 python tools/bench_x3_pc.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
  for v in "" tools/libt3d_noagpr.so; do
    T3D_LIB=$v python bench.py --steps 100 --warmup 20 --no_other_configs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('step', '${v:-agpr(default)}', d['ms_per_step'])"
  done
done
