#!/bin/bash
# diagnostic: the many-tile finalizers -- tests, then the config-4 step with the block shapes of round 2 (T3D_FIN_WIDE=0) and the wide ones
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/fin
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "finaliz" > gpurun_out/fin/tests.log 2>&1; tail -3 gpurun_out/fin/tests.log
for i in 1 2; do
  for v in 0 1024; do
    T3D_FIN_WIDE=$v timeout 300 python bench.py --steps 30 --warmup 5 --batch_size 128 --num_point 2048 --dtype bf16 --no_cpu_baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['per_kernel_us_per_step']; print('bf16 B128 N2048 FIN_WIDE=$v', d['ms_per_step'], d['value'], {n: round(v) for n, v in k.items() if 'finalize' in n})" | tee -a gpurun_out/fin/ab.log
  done
done
