#!/bin/bash
# round-1 measurement set (tag v15) (GPU box): default bench line, single-/two-stream kernel stats, DDP-path line, Large line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/v15
timeout 600 python bench.py > gpurun_out/v15/bench.json 2> gpurun_out/v15/bench.err; tail -1 gpurun_out/v15/bench.json | cut -c1-200
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --force-ddp --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/v15/bench_ddp1.json | cut -c1-200
timeout 300 python bench.py --model large --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/v15/bench_large.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/v15/single" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep metric | cut -c1-160
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/v15/two" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep metric | cut -c1-160
