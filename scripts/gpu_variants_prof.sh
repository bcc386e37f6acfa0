#!/bin/bash
# Build variants of one unit and profile each with rocprofv3 (kernel stats of a 3-step single-stream bench):
#   gpu_variants_prof.sh <unit> <kernel-name-regex> "<flags1>" "<flags2>" ...      ("" = the shipped flags)
u=$1; pat=$2; shift; shift
cd "$GRAFT_REPO_ROOT"
cp hsimae_amd/libhsimae_hip.so /tmp/lib_shipped.so; cp hsimae_amd/build/$u.o /tmp/$u.o.shipped
i=0
for flags in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $flags -c hsimae_amd/csrc/$u.hip -o hsimae_amd/build/$u.o 2>/dev/null || { echo "[$flags] compile failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hsimae_amd/libhsimae_hip.so hsimae_amd/build/*.o
  ms=$(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  d=/tmp/vp_$i; rm -rf $d
  (cd /tmp && TMPDIR=/tmp HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
  echo "[$flags] step $ms ms"
  python3 - "$d" "$pat" <<'P'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f[0])):
        if re.search(sys.argv[2], r["Name"]):
            print("    %-70s calls %4s avg %8.1f us" % (re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
P
done
cp /tmp/$u.o.shipped hsimae_amd/build/$u.o; cp /tmp/lib_shipped.so hsimae_amd/libhsimae_hip.so
