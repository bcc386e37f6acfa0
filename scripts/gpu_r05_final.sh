#!/bin/bash
# end of round 5: full GPU suite, then the measurement set (scripts/gpu_measure_r05.sh)
cd "$GRAFT_REPO_ROOT"; tag=${1:-r05_end}; mkdir -p gpurun_out/$tag
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.txt 2>&1; tail -3 gpurun_out/$tag/pytest_gpu.txt
bash scripts/gpu_measure_r05.sh $tag
