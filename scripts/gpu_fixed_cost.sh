#!/bin/bash
# per-kernel time against the batch size (what part of a launch does not scale with the rows): bash scripts/gpu_fixed_cost.sh
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/fixed; mkdir -p $out; R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
for b in 512 4096; do
  HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/b$b" -- python3 "$R/bench.py" --batch $b --steps 4 --warmup 3 --no-extras 2>&1 | grep -c metric
  cp $R/$out/b$b/*/*_kernel_stats.csv $R/$out/stats_b$b.csv; cp $R/$out/b$b/*/*_kernel_trace.csv $R/$out/trace_b$b.csv
  rm -rf $R/$out/b$b
done
