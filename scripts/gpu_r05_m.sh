#!/bin/bash
# round 5: start stagger of blk256_bwd (larger offsets) and blk256_fwd at Large
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_m; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in base b0 b28 b36 f10 f20; do
  lib=; [ $v != base ] && lib="$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so"
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model large --steps 6 --warmup 2 --no-extras > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_$v/*/*kernel_stats.csv | head -1); grep -E "blk256" $f | cut -d, -f1-4 | sed "s/^/$v /" | sed 's/(anonymous namespace):://g' | cut -c1-150
done | tee $GRAFT_REPO_ROOT/$out/kernels.txt
cd "$GRAFT_REPO_ROOT"
bl() { timeout 300 python bench.py --model large --steps 15 --warmup 5 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
for rep in 1 2 3; do
  echo "large base(b20) $(bl)" >> $out/ab.txt
  for v in b0 b36 f20; do echo "large $v $(HSIMAE_LIB=variants/$v/libhsimae_hip.so bl)" >> $out/ab.txt; done
done
cat $out/ab.txt
