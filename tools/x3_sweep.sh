for cfg in "16384 16384" "8192 16384" "8192 32768" "8192 131072" "1 32768" "16384 32768" "16384 131072" "1 1"; do
  set -- $cfg
  T3D_X3=1 T3D_X3_MINKN=$1 T3D_X3_MINKN_BWD=$2 python bench.py --no_cpu_baseline --no_other_configs --steps 50 --warmup 10 --profile_steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('fwd>=$1 bwd>=$2', round(d['ms_per_step'],4), d['timing']['ms_per_step_min'])"
done
python bench.py --no_cpu_baseline --no_other_configs --steps 50 --warmup 10 --profile_steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('x3 off', round(d['ms_per_step'],4))"
