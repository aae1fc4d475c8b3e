#!/bin/bash
# Development tool: a second library built with extra compiler flags, for an A/B inside one gpurun call (boxes differ by +-4 %):
#   bash tools/build_variant.sh nt -DSEPK_DMA_NT      -> dnn-based_source_separation_amd/libsepkernels_nt.so (git-ignored, travels to the box)
#   SEPKERNELS_LIB=$PWD/dnn-based_source_separation_amd/libsepkernels_nt.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
pkg=dnn-based_source_separation_amd
obj=/tmp/sepk_obj_$name; mkdir -p $obj
for f in $pkg/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude "$@" -c $f -o $obj/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $pkg/libsepkernels_$name.so $obj/*.o
ls -la $pkg/libsepkernels_$name.so
