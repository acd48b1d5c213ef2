#!/bin/bash
# timing ablations of the hand-placed x3 iteration (wrong results, durations only): tools/libt3d_abl_*.so built by tools/build_x3_variant.sh
# with -DT3D_ABL_IL_{NOSTAGE,NOBAR,NOFRAG,NOWRITE,NOLOAD,NOLOAD_R,NOLOAD_C,LOADDUMMY} / -DT3D_ABL_X3_SAMETILE
out=${1:-gpurun_out/il_abl.log}
shapes=${2:-"fwd:512x256 bwd:512x256 bwd:128x128"}
: > $out
for l in transferable3d_amd/libt3d.so tools/libt3d_abl_*.so; do
  for only in $shapes; do
    echo -n "$(basename $l .so | sed 's/libt3d_abl_//') $only  " >> $out
    T3D_LIB=$l T3D_ONLY=$only python tools/bench_x3.py 2>/dev/null | grep -o "x3: .*" | sed 's/x3: .* \([0-9.]* us\).*/\1/' >> $out
  done
done
