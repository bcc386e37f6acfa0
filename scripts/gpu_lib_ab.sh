#!/bin/bash
# Same-box A/B of two builds of the library: bash scripts/gpu_lib_ab.sh <lib A> <lib B> [kernel-name regex]
# (build the reference first, e.g.  git stash; python -m hsimae_amd.build; cp hsimae_amd/libhsimae_hip.so gpurun_out/ab/ref.so; git stash pop)
# kernel stats (single stream, 4 steps) and the two-stream step time, alternating A B A B
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; A=$1; B=$2; pat=${3:-.}
for rep in 1 2; do for L in $A $B; do
  d=/tmp/ab_$RANDOM
  (cd /tmp && TMPDIR=/tmp HSIMAE_LIB="$R/$L" HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$R/bench.py" $HS_BENCH_ARGS --steps 4 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
  echo "== $L"
  python3 - "$d" "$pat" <<'P'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:14]:
    if re.search(sys.argv[2], r["Name"]) and not r["Name"].startswith(("void at::", "__amd")):
        print("    %-52s %8.1f us" % (re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:52], float(r["AverageNs"]) / 1e3))
P
  HSIMAE_LIB="$R/$L" timeout 300 python bench.py $HS_BENCH_ARGS --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('    ms_per_step', d['ms_per_step'])"
done; done
