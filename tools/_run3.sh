cd $GRAFT_REPO_ROOT
for og in 0 1; do
T3D_DP_ONE_GRAPH=$og T3D_FORCE_DIST=1 timeout 300 python bench.py --steps 200 --warmup 20 --no_cpu_baseline --profile_steps 0 2> gpurun_out/og$og.err | tee gpurun_out/og$og.out | wc -l; python -c "import json; d=json.loads(open(\"gpurun_out/og$og.out\").read()); print(\"ONE_GRAPH=$og\", d[\"value\"], d[\"ms_per_step\"], d[\"config\"][\"final_loss\"])"
grep -i "fail\|error" gpurun_out/og$og.err | head -3
done
