#!/bin/bash
# round 4, first call: today's baseline + the half-batch scheduling experiment
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_a; mkdir -p $out
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2> $out/bench_base.err | tail -1 > $out/bench_base.json; cut -c1-300 $out/bench_base.json
timeout 300 python scripts/exp_halfbatch.py > $out/halfbatch.txt 2>&1; grep -v "^img\|^model" $out/halfbatch.txt
HSIMAE_TWO_STREAMS=0 timeout 300 python scripts/exp_halfbatch.py > $out/halfbatch_1s.txt 2>&1; grep -v "^img\|^model" $out/halfbatch_1s.txt
