#!/bin/bash
# Builds a variant of libgsr_hip.so for A/B runs on the GPU box:  tools/mkvariant.sh <name> <file.hip> "<extra -D flags>"
# -> tools/variants/libgsr_<name>.so (git-ignored, travels with gpurun); only <file.hip> is recompiled, the other objects are
# the in-tree build's.  Use with GSR_LIB=$PWD/tools/variants/libgsr_<name>.so (tools/ab_lib.sh).
set -e
name=$1; file=$2; flags=$3
cd "$(dirname "$0")/../gaussian-splatting-slam_amd/csrc"
mkdir -p ../../tools/variants build/var_$name
base=${file%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -fno-slp-vectorize $flags -c $file -o build/var_$name/$base.o
objs=""
for f in api preprocess sort_scan binning render ssim adam densify activations exchange; do
  if [ "$f" = "$base" ]; then objs="$objs build/var_$name/$f.o"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/variants/libgsr_$name.so $objs
echo built tools/variants/libgsr_$name.so
