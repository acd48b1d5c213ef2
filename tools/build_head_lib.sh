#!/bin/bash
# same-box A/B helper: the library of the last commit (git HEAD) -> tools/libt3d_old.so  (select it with T3D_LIB=tools/libt3d_old.so)
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/oldsrc && mkdir -p /tmp/oldsrc
git archive HEAD transferable3d_amd/csrc include | tar -x -C /tmp/oldsrc
( cd /tmp/oldsrc && for f in pointmlp bn_optim fc heads boxpc poolbwd data weak pair; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -w -c transferable3d_amd/csrc/$f.hip -o /tmp/oldsrc/$f.o & done; wait )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libt3d_old.so /tmp/oldsrc/*.o
echo built tools/libt3d_old.so from $(git rev-parse --short HEAD)
