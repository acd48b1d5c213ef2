cd $GRAFT_REPO_ROOT
for v in trace stag16 stag32 stag64; do echo "== $v"; T3D_LIB=tools/libt3d_$v.so timeout 120 python tools/trace_blocks.py 2>&1 | grep -v amdgpu | cut -c1-230 | head -4; done
