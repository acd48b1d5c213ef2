#!/bin/bash
# diagnostic: build libt3d variants with ablation macros into gpurun-visible paths
set -e
cd "$(dirname "$0")/.."
for v in NOSTAGE NOBAR NOEPI "NOSTAGE -DT3D_ABL_NOBAR" "NOSTAGE -DT3D_ABL_NOBAR -DT3D_ABL_NOEPI"; do
  name=$(echo $v | tr -d ' ' | sed 's/-DT3D_ABL_/_/g')
  objs=""
  for f in pointmlp bn_optim fc heads; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DT3D_ABL_$v -c transferable3d_amd/csrc/$f.hip -o /tmp/abl_$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libt3d_$name.so /tmp/abl_pointmlp.o /tmp/abl_bn_optim.o /tmp/abl_fc.o /tmp/abl_heads.o
  echo built tools/libt3d_$name.so
done
