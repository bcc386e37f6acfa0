#!/bin/bash
# fused attention-half backward (blk128_bwd_kernel): parity tests, same-box A/B of the step, kernel stats; wgrad split schedule
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_t}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "fused_attention_half or c2_full or config1 or c1_base48" > $out/pytest_sel.txt 2>&1; tail -2 $out/pytest_sel.txt
HSIMAE_WGRAD_SPLIT=1 timeout 900 python -m pytest tests -m gpu -x -q -k "fused_attention_half_backward or c2_full or config1" > $out/pytest_split.txt 2>&1; tail -2 $out/pytest_split.txt
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2 3; do
  echo "old    $(HSIMAE_FUSED_ATTN_BLOCK_BWD=0 b)" >> $out/ab.txt
  echo "new    $(b)" >> $out/ab.txt
  echo "split  $(HSIMAE_WGRAD_SPLIT=1 b)" >> $out/ab.txt
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
HSIMAE_WGRAD_SPLIT=$v HSIMAE_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | grep -c metric
cp $GRAFT_REPO_ROOT/$out/stats/*/*_kernel_stats.csv $GRAFT_REPO_ROOT/$out/kernel_stats_base_split$v.csv; rm -rf $GRAFT_REPO_ROOT/$out/stats
head -8 $GRAFT_REPO_ROOT/$out/kernel_stats_base_split$v.csv | cut -c1-150
done
