#!/bin/bash
# end of round 6: full GPU suite, then the measurement set (scripts/gpu_measure_r06.sh)
cd "$GRAFT_REPO_ROOT"; tag=${1:-r06_end}; mkdir -p gpurun_out/$tag
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.txt 2>&1; tail -3 gpurun_out/$tag/pytest_gpu.txt
bash scripts/gpu_measure_r06.sh $tag
