#!/bin/bash
# blk256_fwd: parity (fused vs layer-wise, oracle), Large bench A/B, kernel stats
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_e}; out=gpurun_out/$tag; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q -k "d256 or large or wide or adamw or Large or huge_fp8" > $out/pytest_sel.txt 2>&1; tail -4 $out/pytest_sel.txt
grep "fused-attn-half-256" $out/pytest_sel.txt | head
for i in 1 2; do
  HSIMAE_FUSED_ATTN_BLOCK256=0 timeout 300 python bench.py --model large --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | cut -c60-200 | tee -a $out/ab_large_old.txt
  timeout 300 python bench.py --model large --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | cut -c60-200 | tee -a $out/ab_large_new.txt
done
cd /tmp && export TMPDIR=/tmp
HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_large" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model large --steps 3 --warmup 2 --no-extras > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; cp $out/stats_large/*/*_kernel_stats.csv $out/kernel_stats_large_single_stream.csv; rm -rf $out/stats_large
head -14 $out/kernel_stats_large_single_stream.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-120
timeout 300 python bench.py --steps 30 --warmup 8 2>/dev/null | tail -1 > $out/bench_base.json; python -c "
import json; d=json.load(open('$out/bench_base.json')); print(d['ms_per_step'], d['optimizer_step_ms'])"
