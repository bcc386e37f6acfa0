#!/bin/bash
# q|k|v recomputed in the fused attention-half backward: parity, step A/B (saved vs recompute vs separate), kernel stats
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_v}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -s -k "fused_attention_half or c2_full or config1 or tiny or c1_base48 or encode_backward or split_api or dualvit or droppath" > $out/pytest_sel.txt 2>&1; tail -2 $out/pytest_sel.txt; grep "fused-attn-half-bwd" $out/pytest_sel.txt | cut -c1-250
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2 3; do
  echo "separate   $(HSIMAE_FUSED_ATTN_BLOCK_BWD=0 b)" >> $out/ab.txt
  echo "saved      $(HSIMAE_ATTN_BWD_RECOMPUTE=0 b)" >> $out/ab.txt
  echo "recompute  $(b)" >> $out/ab.txt
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
HSIMAE_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | grep -c metric
cp $GRAFT_REPO_ROOT/$out/stats/*/*_kernel_stats.csv $GRAFT_REPO_ROOT/$out/kernel_stats_base.csv; rm -rf $GRAFT_REPO_ROOT/$out/stats
grep blk128_ $GRAFT_REPO_ROOT/$out/kernel_stats_base.csv | cut -c1-150
