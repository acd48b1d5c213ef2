#!/bin/bash
# round 3: rider tests + A/B of the scheduled step against the unscheduled one on one box
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r03/riders_$1
mkdir -p $out
timeout 900 python -m pytest tests/test_riders_gpu.py -x -q -m gpu -s > $out/tests.log 2>&1; tail -5 $out/tests.log
for i in 1 2; do
  T3D_OVERLAP=0 timeout 300 python bench.py --no_cpu_baseline --profile_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('plain    ', d['ms_per_step'], d['value'], d['config']['launches_per_step'])" | tee -a $out/ab.log
  timeout 300 python bench.py --no_cpu_baseline --profile_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('scheduled', d['ms_per_step'], d['value'], d['config']['launches_per_step'], d['config']['schedule'])" | tee -a $out/ab.log
done
timeout 300 python bench.py --no_cpu_baseline --call_detail > $out/bench.json 2> $out/bench.err
grep -v amdgpu.ids $out/bench.err | head -90
