#!/bin/bash
# A/B of environment settings (on the GPU box): usage gpu_ab.sh "ENV1=a ENV2=b" "ENV1=c" ...; REPS alternating repeats each
cd "$GRAFT_REPO_ROOT"
for rep in $(seq 1 ${REPS:-3}); do for cfg in "$@"; do
  echo -n "[$cfg] "
  env $cfg timeout 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | grep metric | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_per_step'], d['roofline'].get('launch_ms'))"
done; done
