#!/bin/bash
# full GPU test suite + bench line (with extras) + decoder kernel stats
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_d}; out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt
timeout 600 python bench.py --steps 40 --warmup 10 2> $out/bench_base.err | tail -1 > $out/bench_base.json; cut -c1-250 $out/bench_base.json
OUTDIR=$out python - <<'PY'
import json
import os
d=json.load(open(os.environ["OUTDIR"]+"/bench_base.json"))
print({k:d.get(k) for k in ("optimizer_step_ms","cpu_baseline")})
print(d["roofline"])
PY
cd /tmp && export TMPDIR=/tmp
HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_new" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-extras > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; cp $out/stats_new/*/*_kernel_stats.csv $out/kernel_stats_base_single_stream.csv; rm -rf $out/stats_new
grep "dec_" $out/kernel_stats_base_single_stream.csv | cut -d, -f1-4 | cut -c1-150
