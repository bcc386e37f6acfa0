#!/bin/bash
# SQ counters per kernel (single stream): bash scripts/gpu_sq.sh <tag>
tag=${1:-sq}
cd /tmp && export TMPDIR=/tmp
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out/$tag"
HSIMAE_TWO_STREAMS=0 timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/$tag" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep metric | cut -c1-120
