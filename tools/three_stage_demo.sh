#!/bin/bash
# The three-stage recipe of the reference's README (stage a: SEMI_MODEL A, stage b: Box-PC Fit net, stage c: SEMI_MODEL F from both
# checkpoints) on synthetic frustums resident in HBM, checkpoints handed over as TensorFlow Saver bundles.  Run on the GPU box:
#   bash tools/three_stage_demo.sh [out_dir]
set -eo pipefail      # a stage that crashes stops the recipe (the next stage would start from missing or stale checkpoints)
out=${1:-gpurun_out/recipe}
mkdir -p $out
# an epoch is one pass over the 16000 frustums (500 steps of 32); checkpoints are written after epochs 0, 5, 10, ...
# (train_semisup.py:316-318): stages a and b train 6 epochs and hand over model_epoch_5.ckpt, i.e. everything that was trained
common="--num_point ${T3D_DEMO_POINTS:-1024} --batch_size 32 --num_channels ${T3D_DEMO_CHANNELS:-4} --device_data 16000 --eval_batches 10 --ckpt_format tf"
python -m transferable3d_amd.train_semisup --SEMI_MODEL A --WEAK_WEIGHT_REPROJECTION 0 --WEAK_WEIGHT_SURFACE 0 $common \
    --max_epoch 6 --log_dir $out/a 2>&1 | grep -v amdgpu > $out/a.log
python -m transferable3d_amd.train_boxpc --BOX_PC_MASK_REPRESENTATION A --BOXPC_WEIGHT_DELTA 4 $common \
    --max_epoch 6 --log_dir $out/b 2>&1 | grep -v amdgpu > $out/b.log
python -m transferable3d_amd.train_semisup_adv --SEMI_MODEL F --BOX_PC_MASK_REPRESENTATION A --use_one_hot --SEMI_TRAIN_BOX_TRAIN_CLASS_AG_TNET 1 \
    --SEMI_TRAIN_BOX_TRAIN_CLASS_AG_BOX 1 --SEMI_BOXPC_FIT_ONLY_ON_2D_CLS 1 --WEAK_WEIGHT_INTRACLASSVAR 2 --WEAK_WEIGHT_REPROJECTION 0 \
    --SEMI_MULTIPLIER_FOR_WEAK_LOSS 0.05 --SEMI_SAMPLE_EQUAL_CLASS_WITH_PROB 1 --SEMI_BOXPC_MIN_FIT_LOSS_AFT_REFINE 1 \
    --SUNRGBD_SEMI_TEST_CLS table sofa dresser night_stand bookshelf \
    --init_class_ag_path $out/a/model_epoch_5.ckpt --init_boxpc_path $out/b/model_epoch_5.ckpt $common \
    --max_epoch 2 --steps_per_epoch 750 --log_dir $out/c 2>&1 | grep -v amdgpu > $out/c.log
grep -E "EPOCH|eval mean|eval box|Mean AP|restored|MEAN|intermediate|refined" $out/a.log $out/b.log $out/c.log
