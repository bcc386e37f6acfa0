#!/bin/bash
# round 5: flakiness check of the final build: smoke(), then the GPU suite twice more on one more box
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_t; mkdir -p $out
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
for i in 1 2; do timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu_$i.txt 2>&1; tail -2 $out/pytest_gpu_$i.txt; done
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; cut -c1-300 $out/bench_default.json
