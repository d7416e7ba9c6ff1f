#!/bin/bash
# Builds a variant of libgsr_hip.so with EVERY source recompiled (for -D flags that live in gsr_common.h):
#   tools/mkvariant_all.sh <name> "<extra -D flags>"   -> tools/variants/libgsr_<name>.so   (use with GSR_LIB=...)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../gaussian-splatting-slam_amd/csrc"
mkdir -p ../../tools/variants build/var_$name
objs=""
for f in api preprocess sort_scan binning render ssim adam densify activations exchange; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -fno-slp-vectorize $flags -c $f.hip -o build/var_$name/$f.o &
  objs="$objs build/var_$name/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/variants/libgsr_$name.so $objs
echo built tools/variants/libgsr_$name.so
