#!/bin/bash
# usage (on the GPU box): bash scripts/gpu_flagsweep.sh "<flags A>" "<flags B>" ...   -- rebuilds the library with each extra hipcc
# flag set (HSIMAE_HIPCC_EXTRA) in a scratch copy of the tree and prints the bench line's ms_per_step
cd "$GRAFT_REPO_ROOT"
for fl in "$@"; do
  rm -rf /tmp/sweep && mkdir -p /tmp/sweep && cp -r hsimae_amd oracle bench.py BASELINE.json include /tmp/sweep/ 2>/dev/null
  ( cd /tmp/sweep && HSIMAE_HIPCC_EXTRA="$fl" python3 -m hsimae_amd.build --force >/dev/null 2>&1 && \
    for r in 1 2; do timeout 100 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline | grep -o "ms_per_step[^,]*," | sed "s|^|[$fl] |"; done )
done
