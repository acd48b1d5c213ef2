# the rider scheduler's cost-model knobs (schedule.py) on the headline step, two passes on one box
for rep in 1 2; do
for e in "" "T3D_RIDER_SLOW=1.3" "T3D_RIDER_SLOW=1.8" "T3D_RIDER_MAXFRAC=0.6" "T3D_RIDER_MAXFRAC=1.5" "T3D_RIDER_STRETCH=0.1" "T3D_RIDER_STRETCH=0.4" "T3D_RIDER_WIDE=0.3" "T3D_RIDER_WIDE=0.8" "T3D_RIDER_WIDE=0" "T3D_RIDER_COST=2.0" "T3D_RIDER_RUN=4"; do
env $e python bench.py --no_cpu_baseline --no_other_configs --steps 150 --warmup 20 --profile_steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); s=d['config'].get('schedule') or {}; print('$e |', round(d['ms_per_step'],4), d['config']['launches_per_step'], s.get('hosted'), s.get('rider_ops'))"
done; done
