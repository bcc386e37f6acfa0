#!/bin/bash
# Several build variants of one unit benchmarked back to back on the GPU box: gpu_variants.sh <unit> "<flags1>" "<flags2>" ...
u=$1; shift
cd "$GRAFT_REPO_ROOT"
cp hsimae_amd/libhsimae_hip.so /tmp/lib_shipped.so; cp hsimae_amd/build/$u.o /tmp/$u.o.shipped
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
echo -n "[shipped] "; run
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $flags -c hsimae_amd/csrc/$u.hip -o hsimae_amd/build/$u.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hsimae_amd/libhsimae_hip.so hsimae_amd/build/*.o
  echo -n "[$flags] "; run
done
cp /tmp/$u.o.shipped hsimae_amd/build/$u.o; cp /tmp/lib_shipped.so hsimae_amd/libhsimae_hip.so
done
