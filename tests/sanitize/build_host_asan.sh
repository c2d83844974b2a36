#!/bin/bash
# TEST INFRASTRUCTURE: builds the library with AddressSanitizer + UBSan on the HOST side only (device code unchanged;
# GPU sanitizers are not available on this pool) into tests/sanitize/_build/libraftcorr_hostasan.so, builds
# host_driver.c against it and runs it.  Needs no GPU: the driver calls the host-only entry points over a sweep of
# shapes and the argument checks of every compute entry point.  A few minutes of hipcc time; run by hand
# (tests/test_sanitizers.py only runs it when RAFTCORR_HOST_ASAN=1), result kept in profiles/r03_sanitizers.txt.
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/.." && cd .. && pwd)
out=$here/_build; mkdir -p $out
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -fno-omit-frame-pointer"
pids=()
for f in api build_pyramid lookup build_bwd bwd_deferred alt_tile alt_fwd16 alt_bwd alt_prep; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $SAN -c $root/raft_amd/csrc/$f.hip -o $out/$f.o &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address -fsanitize=undefined -fno-gpu-sanitize -shared-libsan -o $out/libraftcorr_hostasan.so $out/*.o
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=address -fsanitize=undefined -fno-sanitize-recover=undefined -shared-libsan -I$root/include \
    $here/host_driver.c -o $out/host_driver -L$out -lraftcorr_hostasan -Wl,-rpath,$out -Wl,-rpath,$(dirname $(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so))
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 $out/host_driver
