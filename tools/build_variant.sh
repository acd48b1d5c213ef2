#!/bin/bash
# diagnostic: build libt3d variants:  build_variant.sh name "-DFLAG -DFLAG2" [name2 "flags2" ...]  -> tools/libt3d_<name>.so
set -e
cd "$(dirname "$0")/.."
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  for f in pointmlp pointmlp_x3 bn_optim fc heads boxpc poolbwd data weak pair version; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $flags -c transferable3d_amd/csrc/$f.hip -o /tmp/abl_${name}_$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libt3d_$name.so /tmp/abl_${name}_pointmlp.o /tmp/abl_${name}_pointmlp_x3.o /tmp/abl_${name}_bn_optim.o /tmp/abl_${name}_fc.o /tmp/abl_${name}_heads.o /tmp/abl_${name}_boxpc.o /tmp/abl_${name}_poolbwd.o /tmp/abl_${name}_data.o /tmp/abl_${name}_weak.o /tmp/abl_${name}_pair.o /tmp/abl_${name}_version.o
  echo built tools/libt3d_$name.so
done
