export T3D_PC_MODES=0,1
for v in "" $PC_VARIANTS; do
  echo "== variant: ${v:-base}"
  if [ -n "$v" ]; then export T3D_LIB=tools/libt3d_$v.so; else unset T3D_LIB; fi
  for c in ${PC_CASES:-fwd:512x256 fwd:128x128 fwd:128x1024}; do T3D_ONLY=$c timeout 200 python tools/bench_x3_pc.py 2>&1 | grep -v amdgpu.ids; done
done
