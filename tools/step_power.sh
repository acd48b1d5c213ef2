# clock and package power while the captured step replays back to back (the headline workload), and for the bf16 configuration:
# rocm-smi polled beside the bench process; the samples taken under load (> 500 W) are printed
for cfg in "" "--dtype bf16 --batch_size 128 --num_point 2048"; do
  echo "== bench.py $cfg   (sclk MHz, package W)"
  python bench.py --steps 6000 --warmup 30 --no_other_configs --no_cpu_baseline $cfg > /tmp/step_power_bench.json 2>/dev/null &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower --csv 2>/dev/null | grep card0 | awk -F, '{ if ($NF + 0 > 500) print $6, $NF }'
    sleep 0.3
  done
  wait $pid
  python -c "import json; d=json.loads(open('/tmp/step_power_bench.json').read().strip().split(chr(10))[-1]); print('ms_per_step', d['ms_per_step'])"
done
