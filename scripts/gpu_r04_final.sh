#!/bin/bash
# end-of-round: full GPU suite, then the measurement set
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_y}; mkdir -p gpurun_out/$tag
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.txt 2>&1; tail -3 gpurun_out/$tag/pytest_gpu.txt
bash scripts/gpu_measure_r04.sh $tag
