#!/bin/bash
# round 5: planar weight-gradient operands (new default) against HSIMAE_WGRAD_PLANAR=0; delta-in-accumulator A/B; parity
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_e; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sched.py tests/test_gpu_e2e.py -m gpu -x -q -k "planar or wgrad or mlp or sched or switch or c1_base48 or c2_full or tiny or large" > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
b() { timeout 300 python bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
for rep in 1 2 3; do
  echo "base(planar+delta_in_acc) $(b)" >> $out/ab.txt
  echo "row-major operands        $(HSIMAE_WGRAD_PLANAR=0 b)" >> $out/ab.txt
  echo "delta by v_sub            $(HSIMAE_LIB=variants/nodelta/libhsimae_hip.so b)" >> $out/ab.txt
done
cat $out/ab.txt
for m in large; do echo "$m planar $(timeout 300 python bench.py --model $m --steps 15 --warmup 5 --no-extras 2>/dev/null | tail -1 | cut -c60-200)"; echo "$m rowmajor $(HSIMAE_WGRAD_PLANAR=0 timeout 300 python bench.py --model $m --steps 15 --warmup 5 --no-extras 2>/dev/null | tail -1 | cut -c60-200)"; done | tee $out/large.txt
cd /tmp && export TMPDIR=/tmp
for v in base rowmajor nodelta; do
  lib=; pl=1; [ $v = nodelta ] && lib="$GRAFT_REPO_ROOT/variants/nodelta/libhsimae_hip.so"; [ $v = rowmajor ] && pl=0
  HSIMAE_WGRAD_PLANAR=$pl HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-extras > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_$v/*/*kernel_stats.csv | head -1); head -7 $f | cut -d, -f1-4 | sed "s/^/$v /" | cut -c1-200
done
