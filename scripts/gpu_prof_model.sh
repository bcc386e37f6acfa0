#!/bin/bash
# usage (GPU box): gpu_prof_model.sh <tag> <model> [bench args...] -- single-stream rocprofv3 kernel stats of a short bench run
tag=$1; model=$2; shift 2
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp
export HSIMAE_TWO_STREAMS=0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/$tag" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $model --steps 3 --warmup 2 --no-extras "$@" 2>&1 | grep metric | cut -c1-200
cd "$GRAFT_REPO_ROOT"; python scripts/prof_summary.py gpurun_out/$tag 2>/dev/null | head -24
