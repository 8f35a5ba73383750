#!/bin/bash
# Build a variant of libmvo_hip.so with extra -D flags (kernel experiments; the product build is csrc/Makefile).
#   tools/build_variant.sh NAME -DFOO=1 -DBAR=2   ->  build/libmvo_NAME.so   (select it with MVO_LIB=build/libmvo_NAME.so)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/obj_$name
mkdir -p $out
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Xclang -target-feature -Xclang -unaligned-access-mode"
pids=()
for f in $root/ros2_mono_vo_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  extra=""; [ $b = match ] && extra="-mllvm -amdgpu-mfma-vgpr-form"   # as csrc/Makefile
  /opt/rocm/bin/hipcc $FLAGS $extra "$@" -c $f -o $out/$b.o 2> $out/$b.log &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/build/libmvo_$name.so $out/*.o -lpthread -ldl
echo built $root/build/libmvo_$name.so
