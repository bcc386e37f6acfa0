#!/bin/bash
# rocprofv3 kernel stats of the Large (D = 256) configuration on the GPU box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/large; cd /tmp && export TMPDIR=/tmp
HSIMAE_TWO_STREAMS=${TS:-1} timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/large" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model large --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep metric | cut -c1-160
cd "$GRAFT_REPO_ROOT"; python scripts/prof_summary.py gpurun_out/large 5 16
