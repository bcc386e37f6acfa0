#!/bin/bash
# HBM traffic counters for the bench kernels (separate passes, no other trace domains): bash scripts/gpu_pmc.sh <tag>
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p "$GRAFT_REPO_ROOT/gpurun_out/$tag/$c"
  timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/$tag/$c" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep metric | cut -c1-120
done
ls -R "$GRAFT_REPO_ROOT/gpurun_out/$tag" | head -20
