#!/bin/bash
# round 5: blk256_bwd_kernel knobs (epilogue rows requested early, start stagger), kernel stats at Large
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in base ex s6 s12 exs12 s20; do
  lib=; [ $v != base ] && lib="$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so"
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model large --steps 6 --warmup 2 --no-extras > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_$v/*/*kernel_stats.csv | head -1); grep -E "blk256" $f | cut -d, -f1-4 | sed "s/^/$v /" | sed 's/(anonymous namespace):://g' | cut -c1-150
done | tee $GRAFT_REPO_ROOT/$out/kernels.txt
cd "$GRAFT_REPO_ROOT"
bl() { timeout 300 python bench.py --model large --steps 15 --warmup 5 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
for rep in 1 2; do
  echo "large base $(bl)" >> $out/ab.txt
  for v in ex s12 exs12; do echo "large $v $(HSIMAE_LIB=variants/$v/libhsimae_hip.so bl)" >> $out/ab.txt; done
done
cat $out/ab.txt
