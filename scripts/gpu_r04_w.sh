#!/bin/bash
# blk128_bwd_kernel LDS layout: unpadded swizzled images (default) vs the 272-byte pitch (variants/oldlds)
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_w}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "fused_attention_half or c2_full or config1 or tiny or c1_base48" > $out/pytest_sel.txt 2>&1; tail -2 $out/pytest_sel.txt
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2 3; do
  echo "old    $(HSIMAE_LIB=$PWD/variants/oldlds/libhsimae_hip.so b)" >> $out/ab.txt
  echo "new    $(b)" >> $out/ab.txt
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in oldlds default; do
if [ $v = oldlds ]; then export HSIMAE_LIB=$GRAFT_REPO_ROOT/variants/oldlds/libhsimae_hip.so; else unset HSIMAE_LIB; fi
HSIMAE_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | grep -c metric
cp $GRAFT_REPO_ROOT/$out/stats/*/*_kernel_stats.csv $GRAFT_REPO_ROOT/$out/kernel_stats_base_$v.csv; rm -rf $GRAFT_REPO_ROOT/$out/stats
grep blk128_ $GRAFT_REPO_ROOT/$out/kernel_stats_base_$v.csv | cut -c1-150
done
