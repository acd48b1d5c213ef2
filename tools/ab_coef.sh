timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
echo "== coef LDS (default)"; timeout 300 python tools/bench_x3.py 2>&1 | grep -v amdgpu.ids | sed 's/fp32-MFMA.*x3:/x3:/'
echo "== coef global (variant)"; T3D_LIB=tools/libt3d_nocoef.so timeout 300 python tools/bench_x3.py 2>&1 | grep -v amdgpu.ids | sed 's/fp32-MFMA.*x3:/x3:/'
for i in 1 2 3; do for v in "" tools/libt3d_nocoef.so; do
T3D_LIB=$v python bench.py --steps 200 --warmup 30 --no_other_configs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('step', '${v:-default}', d['ms_per_step'])"
done; done
