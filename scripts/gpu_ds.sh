#!/bin/bash
# wgrad ring-depth / workgroup-budget sweep (on the GPU box): step ms and wgrad launch ms per setting
cd "$GRAFT_REPO_ROOT"
for ds in ${DSS:-3 4 5 6}; do for wgs in ${WGSS:-128 256 384 512}; do
  echo -n "DS=$ds WGS=$wgs: "
  HSIMAE_WGRAD_DS=$ds HSIMAE_WGRAD_WGS=$wgs timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_per_step'], d['roofline'].get('launch_ms'))"
done; done
