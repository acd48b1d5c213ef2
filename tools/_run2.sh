cd $GRAFT_REPO_ROOT
for v in 0 1; do
  T3D_FC_SIDE=$v timeout 200 python bench.py --no_cpu_baseline --profile_steps 0 --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FC_SIDE=$v graph', d['value'], d['ms_per_step'])"
  T3D_FC_SIDE=$v timeout 200 python bench.py --no_cpu_baseline --profile_steps 0 --steps 200 --no_graph 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FC_SIDE=$v eager', d['value'], d['ms_per_step'])"
done
T3D_FC_SIDE=1 timeout 600 python -m pytest tests/test_model_gpu.py tests/test_eval_gpu.py -m gpu -x -q 2>&1 | tail -3
