#!/bin/bash
# round 5: operand-layout micro-benchmark (fixed), full-size oracle parity tests (gate calibration)
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_b; mkdir -p $out
timeout 120 ./scripts/micro/hbm_stride.bin > $out/hbm_stride.txt 2>&1; cat $out/hbm_stride.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s > $out/pytest_fullsize.txt 2>&1; grep -E "^\[|passed|failed|Error|assert" $out/pytest_fullsize.txt | cut -c1-400 | tail -20
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_boundary.py -m gpu -x -q -k "full_size or middle_gradient" > $out/pytest_b.txt 2>&1; tail -3 $out/pytest_b.txt
