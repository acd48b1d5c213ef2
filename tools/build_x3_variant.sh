#!/bin/bash
# build tools/libt3d_<name>.so = the current objects with pointmlp_x3 recompiled under extra flags (fast: the x3 TU only)
set -e
cd /root/repo
name=$1; flags=$2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $flags -c transferable3d_amd/csrc/pointmlp_x3.hip -o /tmp/x3v_$name.o 2>/tmp/x3v_$name.log || { cat /tmp/x3v_$name.log; exit 1; }
objs=$(ls transferable3d_amd/build/*.o | grep -v pointmlp_x3)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libt3d_$name.so $objs /tmp/x3v_$name.o
echo built tools/libt3d_$name.so
