#!/bin/bash
# Builds a lab variant of the library: tools/build_variant.sh NAME -DFLAG ... -> tools/libraftcorr_NAME.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/rc_var_$name; mkdir -p $obj
for f in api build_pyramid lookup build_bwd bwd_deferred alt_tile alt_fwd16 alt_bwd alt_prep; do
  [ -f $root/raft_amd/csrc/$f.hip ] || continue
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DRC_LAB "$@" -c $root/raft_amd/csrc/$f.hip -o $obj/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tools/libraftcorr_$name.so $obj/*.o
echo built tools/libraftcorr_$name.so
