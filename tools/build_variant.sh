#!/bin/bash
# usage: tools/build_variant.sh NAME "-DPT_X=.. -DPT_Y=.."   -> pytracking_amd/variants/libpt_hot_NAME.so (experiments only;
# select with PT_HOT_LIB=...).  Only fast_passes.hip is recompiled with the extra flags; the other objects come from build/.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../pytracking_amd"
mkdir -p variants build
python -c "import sys; sys.path.insert(0, '..'); from pytracking_amd import _lib; _lib.build_library()"
SRC=${PT_VARIANT_SRC:-fast_passes}
OBJS=""
for o in build/*.o; do b=$(basename $o .o); [[ " $SRC " == *" $b "* ]] || OBJS="$OBJS $o"; done
for s in $SRC; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=14 $@ -c csrc/$s.hip -o variants/${s}_$NAME.o
  OBJS="$OBJS variants/${s}_$NAME.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libpt_hot_$NAME.so $OBJS
echo built variants/libpt_hot_$NAME.so
