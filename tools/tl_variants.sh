#!/bin/bash
# ON THE GPU BOX: step timelines (tools/step_timeline.py) of bench.py under several environments.
#   bash tools/tl_variants.sh <outdir> "<bench flags>" "name:ENV=1 ENV2=2:extra flags" ...
out=$1; flags=$2; shift 2
root=$(pwd); mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%:*}; rest=${v#*:}; envs=${rest%%:*}; extra=${rest#*:}
  [ "$extra" = "$rest" ] && extra=""
  rm -rf /tmp/tl_$name
  ( for e in $envs; do export $e; done
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$name -o run -- python3 $root/bench.py $flags $extra --no_cpu_baseline --no_other_configs --steps 20 --warmup 5 --profile_steps 0 > $root/$out/$name.json 2> $root/$out/$name.err )
  f=$(find /tmp/tl_$name -name "*kernel_trace.csv" | head -1)
  python3 $root/tools/step_timeline.py $f > $root/$out/$name.txt 2>&1
  echo "$name: $(python3 -c "import json,sys; d=json.load(open('$root/$out/$name.json')); print(d['ms_per_step'])" 2>&1 | tail -1) ms;  $(head -1 $root/$out/$name.txt)"
done
