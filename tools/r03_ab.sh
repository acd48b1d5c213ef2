#!/bin/bash
# round 3: same-box A/B of step variants selected by environment variables:  bash tools/r03_ab.sh <tag> "VAR=.. VAR=.." "VAR=.." ...
cd "$(dirname "$0")/.." || exit 1
tag=$1; shift
out=gpurun_out/r03/ab_$tag
mkdir -p $out
for rep in 1 2; do
  for v in "$@"; do
    env $v timeout 300 python bench.py --no_cpu_baseline --no_other_configs --profile_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-44s' % '$v', round(d['ms_per_step'],4), round(d['value']), d['config']['launches_per_step'])" | tee -a $out/ab.log
  done
done
