#!/bin/bash
# round 5: second stagger sweep: dec_bwd_attn step size at 2 classes, + dec_bwd_mlp small steps, + blk128_bwd / blk128_fwd
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_i; mkdir -p $out
b() { timeout 300 python bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
b > /dev/null
for rep in 1 2 3; do
  echo "base  $(b)" >> $out/ab.txt
  for v in a2_2 a2_4 a2_6 a2_8 a2_6m2 a2_6b3 a2_6b6f3 split a2_6split; do echo "$v $(HSIMAE_LIB=variants/$v/libhsimae_hip.so b)" >> $out/ab.txt; done
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in base a2_2 a2_4 a2_6 a2_8 a2_6m2 a2_6b3 a2_6b6f3 split a2_6split; do
  lib=; [ $v != base ] && lib="$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so"
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-extras > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_$v/*/*kernel_stats.csv | head -1); grep -E "dec_bwd|blk128" $f | cut -d, -f1-4 | sed "s/^/$v /" | sed 's/(anonymous namespace):://g' | cut -c1-120
done
