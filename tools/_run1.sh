set -x
cd $GRAFT_REPO_ROOT
T3D_FORCE_DIST=1 timeout 300 python bench.py --steps 50 --warmup 10 --no_cpu_baseline --profile_steps 0 > gpurun_out/dist1.json 2> gpurun_out/dist1.err; echo rc=$?
tail -3 gpurun_out/dist1.err; cat gpurun_out/dist1.json
T3D_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 10 --no_cpu_baseline --profile_steps 0 > gpurun_out/dist2.json 2> gpurun_out/dist2.err; echo rc=$?
tail -3 gpurun_out/dist2.err; cat gpurun_out/dist2.json
timeout 200 python tools/bench_gemm.py fwd > gpurun_out/gemm_fwd_m32k.txt 2>&1
T3D_M=262144 timeout 200 python tools/bench_gemm.py fwd > gpurun_out/gemm_fwd_m256k.txt 2>&1
timeout 200 python tools/bench_gemm.py dgrad > gpurun_out/gemm_dgrad_m32k.txt 2>&1
T3D_M=262144 timeout 200 python tools/bench_gemm.py dgrad > gpurun_out/gemm_dgrad_m256k.txt 2>&1
paste gpurun_out/gemm_fwd_m32k.txt gpurun_out/gemm_fwd_m256k.txt | cut -c1-220
