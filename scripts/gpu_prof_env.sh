#!/bin/bash
# usage (GPU box): gpu_prof_env.sh <tag> [ENV=val ...] -- rocprofv3 kernel stats of a short bench run under the given environment
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/$tag" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep metric | cut -c1-120
cd "$GRAFT_REPO_ROOT"; python scripts/prof_summary.py gpurun_out/$tag 2>/dev/null | head -16
