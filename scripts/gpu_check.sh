#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/gpu_check.sh <tag> [pytest -k expr]
# runs a subset of the GPU tests, then a short rocprofv3 kernel trace of bench.py into gpurun_out/<tag>/
tag=${1:-prof}; kexpr=${2:-"fused or c1_base"}
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_e2e.py -m gpu -q -s --timeout 600 -k "$kexpr" 2>&1 | grep "^\[\|passed\|failed\|Error" | tail -12
mkdir -p gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/$tag" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep metric | cut -c1-220
