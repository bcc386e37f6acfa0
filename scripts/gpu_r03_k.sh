#!/bin/bash
tag=${1:-r03_k}
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for cfg in "base bf16" "base fp8" "large bf16" "large fp8" "huge fp8" "huge bf16"; do
  set -- $cfg
  timeout 300 python bench.py --model $1 --precision $2 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $out/bench_$1_$2.json
  python -c "import json,sys; d=json.load(open('$out/bench_$1_$2.json')); print('$1 $2', d['ms_per_step'], d['value'])"
done
