#!/bin/bash
# quick look at a kernel change on the GPU box: bash scripts/gpu_try.sh "<pytest -k expression>" "<kernel-name regex>"
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q -k "${1:-decoder}" 2>&1 | tail -3
d=/tmp/try_$RANDOM
(cd /tmp && TMPDIR=/tmp HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$R/bench.py" $HS_BENCH_ARGS --steps 4 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
python3 - "$d" "${2:-.}" <<'P'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:18]:
    if re.search(sys.argv[2], r["Name"]) and not r["Name"].startswith(("void at::", "__amd")):
        print("    %-60s calls %4s avg %8.1f us" % (re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
P
for i in 1 2; do timeout 300 python bench.py $HS_BENCH_ARGS --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"; done
