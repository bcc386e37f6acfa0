#!/bin/bash
# round 5: blk256_bwd_kernel (Large: the attention half's backward as one launch): parity against the three launches it replaces, Large A/B, kernel stats
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_k; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_sched.py tests/test_gpu_boundary.py -m gpu -x -q -s -k "d256 or large or Large or BLOCK256 or c3" > $out/pytest.txt 2>&1; grep -E "^\[fused-attn-half-256 bwd|passed|failed|Error|assert" $out/pytest.txt | cut -c1-300 | tail -20
bl() { timeout 300 python bench.py --model large --steps 15 --warmup 5 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
for rep in 1 2 3; do
  echo "large blk256_bwd      $(bl)" >> $out/ab.txt
  echo "large three launches  $(HSIMAE_FUSED_ATTN_BLOCK256_BWD=0 bl)" >> $out/ab.txt
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in fused separate; do
  e=1; [ $v = separate ] && e=0
  HSIMAE_FUSED_ATTN_BLOCK256_BWD=$e HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model large --steps 6 --warmup 2 --no-extras > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_$v/*/*kernel_stats.csv | head -1); head -14 $f | cut -d, -f1-4 | sed "s/^/$v /" | sed 's/(anonymous namespace):://g' | cut -c1-150
done
