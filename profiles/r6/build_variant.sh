#!/bin/bash
# Builds an experimental copy of the library: bash profiles/r6/build_variant.sh <name> <extra hipcc flags...>
# -> diff-gaussian-rasterization_amd/lib/libdgr_hip_<name>.so (objects in build_<name>/); select it with DGR_HIP_LIB.
set -e
cd "$(dirname "$0")/../../diff-gaussian-rasterization_amd"
N=$1; shift
mkdir -p build_$N lib
for f in api preprocess binning segment_binning render_light render_full optim slam; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function "$@" -c csrc/$f.hip -o build_$N/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -o lib/libdgr_hip_$N.so build_$N/*.o
rm -rf build_$N
ls -la lib/libdgr_hip_$N.so
