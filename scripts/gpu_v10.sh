#!/bin/bash
# round-1 v10 measurement set (GPU box): default bench line, single- and two-stream kernel stats, Large config line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/v10
timeout 600 python bench.py > gpurun_out/v10/bench.json 2> gpurun_out/v10/bench.err; tail -1 gpurun_out/v10/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/v10/single" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep metric | cut -c1-160
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/v10/two" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep metric | cut -c1-160
cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --model large --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300 | tee gpurun_out/v10/bench_large.json
