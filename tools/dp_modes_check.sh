pr() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(d['value']), round(d['ms_per_step'],4), d['config'].get('graph_segments_per_step'), json.dumps(d['config'].get('dp',{}).get('modes')), d['config'].get('dp',{}).get('headline_mode'), d.get('hang'))"; }
T3D_FORCE_DIST=1 python bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_other_configs 2>/dev/null | pr default_one_rank
T3D_FORCE_DIST=1 T3D_DP_SAFE_FIRST=2 python bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_other_configs 2>/dev/null | pr safe_first_one_rank
T3D_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | pr gloo_two_ranks_one_gpu
python bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_other_configs 2>/dev/null | pr single
