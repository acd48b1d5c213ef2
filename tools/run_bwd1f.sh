#!/bin/bash
# diagnostic: the fp32 one-pass backward -- kernel tests, per-shape timings against the split form, phase trace, step A/B on one box
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/bwd1f
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "one_pass or fused_bwd" > gpurun_out/bwd1f/tests.log 2>&1
tail -3 gpurun_out/bwd1f/tests.log
for kn in "128 128" "128 64" "64 128" "64 64"; do
  for M in 32768 65536 131072 262144; do
    timeout 120 python tools/bench_bwd_bf16.py $kn $M 20 0 f32 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bwd1f/shapes.log
  done
done
if [ -f tools/libt3d_trace.so ]; then
  T3D_LIB=tools/libt3d_trace.so timeout 200 python tools/trace_bwd1f.py 262144 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bwd1f/trace.log
  T3D_LIB=tools/libt3d_trace.so timeout 200 python tools/trace_bwd1f.py 32768 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bwd1f/trace.log
fi
if [ "$1" = "ab" ]; then
for i in 1 2; do
  T3D_BWD1F=0 timeout 300 python bench.py --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('split   ', d['ms_per_step'], d['value'])" | tee -a gpurun_out/bwd1f/ab.log
  timeout 300 python bench.py --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('one-pass', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'])" | tee -a gpurun_out/bwd1f/ab.log
done
fi
