#!/bin/bash
# round 5, first call: operand-layout micro-benchmark, full GPU suite on the schedule-record refactor, baseline bench lines
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_a; mkdir -p $out
./scripts/micro/hbm_stride.bin > $out/hbm_stride.txt 2>&1; cat $out/hbm_stride.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -5 $out/pytest_gpu.txt
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-330; done > $out/bench_base.txt; cat $out/bench_base.txt
