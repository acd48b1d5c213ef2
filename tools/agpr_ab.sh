# same-box A/B of the accumulator placement of the x3 kernels (T3D_X3_AGPR): per-shape launch times, then the whole step.
# The committed default is T3D_X3_AGPR=0 (accumulators in VGPRs); the variant is built with
#   bash tools/build_x3_variant.sh agpr "-DT3D_X3_AGPR=1"      ->  tools/libt3d_agpr.so
# (round 5's version of this script had the two labels the wrong way round: profiles/r05_x3_agpr_ab.log reads "AGPR (default build)" for
# what was the VGPR default and "VGPR" for the AGPR variant)
export T3D_PC_MODES=0
[ -f tools/libt3d_agpr.so ] || bash tools/build_x3_variant.sh agpr "-DT3D_X3_AGPR=1"
echo "== VGPR accumulators (default build)"; timeout 300 python tools/bench_x3_pc.py 2>&1 | grep -v amdgpu.ids
echo "== AGPR accumulators (-DT3D_X3_AGPR=1, tools/libt3d_agpr.so)"; T3D_LIB=tools/libt3d_agpr.so timeout 300 python tools/bench_x3_pc.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
  for v in "" tools/libt3d_agpr.so; do
    T3D_LIB=$v python bench.py --steps 100 --warmup 20 --no_other_configs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('step', '${v:-vgpr(default)}', d['ms_per_step'])"
  done
done
