#!/bin/bash
# MFMA busy counters per kernel (single stream): bash scripts/gpu_mfma.sh <tag>
tag=${1:-mfma}
cd /tmp && export TMPDIR=/tmp
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out/$tag"
HSIMAE_TWO_STREAMS=0 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/$tag" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -i "metric\|error\|invalid" | cut -c1-160
