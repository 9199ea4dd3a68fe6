#!/bin/bash
# Build libunicorn_hip.so for gfx950 (MI355X) + the host-side association library.  hipcc cross-compiles without a GPU.
# Incremental: a translation unit is rebuilt when it is older than its source or than ANY header (*.h here, include/*.h) -- a glob, not a list.
# Writes ../lib/build_manifest.json (sha256 of every source / header -> the two libraries); unicorn_amd/_lib.py verifies it at load, so a
# library that was not built from the sources next to it fails loudly instead of running stale code on the GPU box.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
UNITS="gemm gemm_h2 gemm_h2d gemm_h2q gemm_p44 mlp_fused norm msda corr misc post mask_post engine api"
HDRS=$(ls *.h ../../include/*.h)
pids=()
objs=()
for f in $UNITS; do
  objs+=(build/$f.o)
  stale=0
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ]; then stale=1; fi
  for h in $HDRS; do if [ $h -nt build/$f.o ]; then stale=1; fi; done
  if [ $stale = 1 ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o $OUT/libunicorn_hip.so
echo "built $OUT/libunicorn_hip.so"
# host-side association library (row N2): plain C++, no HIP
g++ -O3 -std=c++17 -fPIC -shared -ffp-contract=off -o $OUT/libunicorn_assoc.so assoc.cpp
echo "built $OUT/libunicorn_assoc.so"
python3 - "$OUT" $UNITS <<'EOF'
import hashlib, json, os, sys, glob
out, units = sys.argv[1], sys.argv[2:]
def sha(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()
srcs = sorted([u + ".hip" for u in units] + ["assoc.cpp"] + glob.glob("*.h") + glob.glob("../../include/*.h"))
man = {"sources": {os.path.basename(p): sha(p) for p in srcs},
       "libs": {n: sha(os.path.join(out, n)) for n in ("libunicorn_hip.so", "libunicorn_assoc.so")}}
json.dump(man, open(os.path.join(out, "build_manifest.json"), "w"), indent=1, sort_keys=True)
print("wrote %s/build_manifest.json (%d sources)" % (out, len(srcs)))
EOF
# standalone measurement program (tools/feed_probe.hip: which path feeds a CU; profiles/r05_feed_probe.txt) -- not part of the library:
# built last and non-fatally (a probe that does not compile on another ROCm must not leave the libraries unbuilt)
mkdir -p ../../tools/build
for probe in feed_probe persist_probe winograd_probe handoff_probe; do      # persist_probe: kernel boundary vs grid barrier; winograd_probe: F(2x2,3x3) transform passes (profiles/r06_*)
  if [ ! -f ../../tools/build/$probe ] || [ ../../tools/$probe.hip -nt ../../tools/build/$probe ]; then
    hipcc --offload-arch=gfx950 -O3 -Wno-unused-value ../../tools/$probe.hip -o ../../tools/build/$probe || echo "$probe skipped (does not build here)"
  fi
done
# a host without Python / torch (tools/capi_host_demo.cpp: the SOT step through include/unicorn_hip.h + the HIP runtime only), non-fatal like the probes
if [ ! -f ../../tools/build/capi_host_demo ] || [ ../../tools/capi_host_demo.cpp -nt ../../tools/build/capi_host_demo ] || [ ../../include/unicorn_hip.h -nt ../../tools/build/capi_host_demo ]; then
  hipcc -O2 -std=c++17 ../../tools/capi_host_demo.cpp -L$OUT -lunicorn_hip -Wl,-rpath,'$ORIGIN/../../unicorn_amd/lib' -o ../../tools/build/capi_host_demo || echo "capi_host_demo skipped (does not build here)"
fi

