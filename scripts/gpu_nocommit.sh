#!/bin/bash
# Timing experiment (results wrong by construction): every gradient commit (hs_gadd) compiled out in the given units, kernel
# stats before / after:   bash scripts/gpu_nocommit.sh "<unit> <unit> ..."  [kernel-name regex]
units=${1:-"wgrad fused_dec fused_enc gemm gemm_dma elem"}; pat=${2:-"."}
cd "$GRAFT_REPO_ROOT"
cp hsimae_amd/libhsimae_hip.so /tmp/lib_shipped.so
for flags in "" "-DHS_EXP_NO_COMMIT"; do
  objs=""
  for u in gemm gemm_dma attn wgrad elem pack fused_dec fused_enc loader api; do
    if [[ " $units " == *" $u "* ]]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $flags -c hsimae_amd/csrc/$u.hip -o /tmp/$u.o 2>/dev/null; objs="$objs /tmp/$u.o"
    else objs="$objs hsimae_amd/build/$u.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hsimae_amd/libhsimae_hip.so $objs
  d=/tmp/nc_$RANDOM
  (cd /tmp && TMPDIR=/tmp HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$GRAFT_REPO_ROOT/bench.py" $HS_BENCH_ARGS --steps 3 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
  echo "[$flags]"; python3 - "$d" "$pat" <<'P'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:16]:
    if re.search(sys.argv[2], r["Name"]) and not r["Name"].startswith(("void at::", "__amd")):
        print("    %-60s calls %4s avg %8.1f us" % (re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
P
done
cp /tmp/lib_shipped.so hsimae_amd/libhsimae_hip.so
