#!/bin/bash
# Build libunicorn_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
pids=()
for f in gemm gemm_h2 gemm_h2d gemm_h2q gemm_p44 mlp_fused norm msda corr misc post mask_post engine api; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ kernels.h -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ engine.h -nt build/$f.o ] || [ gemm_epi.h -nt build/$f.o ] || [ mask_interp.h -nt build/$f.o ] || [ ../../include/unicorn_hip.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/gemm.o build/gemm_h2.o build/gemm_h2d.o build/gemm_h2q.o build/gemm_p44.o build/mlp_fused.o build/norm.o build/msda.o build/corr.o build/misc.o build/post.o build/mask_post.o build/engine.o build/api.o -o $OUT/libunicorn_hip.so
echo "built $OUT/libunicorn_hip.so"
# standalone measurement program (tools/feed_probe.hip: which path feeds a CU; profiles/r05_feed_probe.txt) -- not part of the library
mkdir -p ../../tools/build
if [ ! -f ../../tools/build/feed_probe ] || [ ../../tools/feed_probe.hip -nt ../../tools/build/feed_probe ]; then
  hipcc --offload-arch=gfx950 -O3 -Wno-unused-value ../../tools/feed_probe.hip -o ../../tools/build/feed_probe
fi
# host-side association library (row N2): plain C++, no HIP
g++ -O3 -std=c++17 -fPIC -shared -ffp-contract=off -o $OUT/libunicorn_assoc.so assoc.cpp
echo "built $OUT/libunicorn_assoc.so"
