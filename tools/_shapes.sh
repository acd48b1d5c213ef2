cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for cfg in "16 512 6" "64 1024 3" "8 128 4" "128 256 6" "1 128 4"; do set -- $cfg
  echo "== B=$1 N=$2 C=$3"
  timeout 200 python -m transferable3d_amd.train_semisup --SEMI_MODEL A --WEAK_WEIGHT_REPROJECTION 0 --WEAK_WEIGHT_SURFACE 0 --num_point $2 --batch_size $1 --num_channels $3 --max_epoch 1 --steps_per_epoch 30 --device_data 300 --eval_batches 2 --log_dir gpurun_out/shapes 2>&1 | grep -E "EPOCH 000 \*|Error|error|eval mean" | head -4
done
timeout 200 python bench.py --no_cpu_baseline --profile_steps 0 2>/dev/null | cut -c1-120
