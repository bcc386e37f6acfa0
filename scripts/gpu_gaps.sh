#!/bin/bash
# idle time between kernels of the base step, one and two streams: bash scripts/gpu_gaps.sh
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/gaps; mkdir -p $out; R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
for ts in 0 1; do
  HSIMAE_TWO_STREAMS=$ts timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$out/t$ts" -- python3 "$R/bench.py" --steps 4 --warmup 3 --no-extras 2>&1 | grep -c metric
  f=$(ls $R/$out/t$ts/*/*_kernel_trace.csv | head -1); echo "== two_streams=$ts"; python3 $R/scripts/trace_gaps.py $f 3
done
