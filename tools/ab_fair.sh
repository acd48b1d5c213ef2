# same-box A/B of T3D_X3_FAIR (the younger workgroup of a CU pair leads the first part of its k loop at wave priority 1):
# per-workgroup timeline, per-shape launch times, whole step.  Variants: tools/build_x3_variant.sh nofair "-DT3D_X3_FAIR=0" etc.
echo "== timeline, default build (T3D_X3_FAIR=1, 1/2)"
T3D_X3=1 T3D_X3_MINKN=1 T3D_TRACE_STRIDE=8 T3D_LIB=tools/libt3d_trace8.so timeout 600 python tools/trace_blocks.py 2>&1 | grep -v amdgpu.ids
for v in "" tools/libt3d_nofair.so tools/libt3d_fair13.so tools/libt3d_fair23.so; do
  echo "== launches ${v:-default (fair 1/2)}"; T3D_LIB=$v timeout 300 python tools/bench_x3.py 2>&1 | grep -v amdgpu.ids | sed 's/fp32-MFMA.*x3:/x3:/'
done
for i in 1 2 3; do for v in "" tools/libt3d_nofair.so tools/libt3d_fair13.so tools/libt3d_fair23.so; do
T3D_LIB=$v python bench.py --steps 200 --warmup 30 --no_other_configs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('step', '${v:-default}', d['ms_per_step'])"
done; done
