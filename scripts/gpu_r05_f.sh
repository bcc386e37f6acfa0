#!/bin/bash
# round 5: plane stride of the planar operands (M + 48 rows, default, against M and M + 16), planar test, finer phase stamps in dec_bwd_attn
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_f; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sched.py tests/test_gpu_e2e.py -m gpu -x -q -k "planar or wgrad or mlp or PLANAR or attention_half or decoder or c1_base48" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
b() { timeout 300 python bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
for rep in 1 2 3; do
  echo "pad48(default) $(b)" >> $out/ab.txt
  echo "pad0           $(HSIMAE_LIB=variants/pad0/libhsimae_hip.so b)" >> $out/ab.txt
  echo "pad16          $(HSIMAE_LIB=variants/pad16/libhsimae_hip.so b)" >> $out/ab.txt
  echo "row-major      $(HSIMAE_WGRAD_PLANAR=0 b)" >> $out/ab.txt
  echo "blk128_fwd 272-byte pitch $(HSIMAE_LIB=variants/noswz/libhsimae_hip.so b)" >> $out/ab.txt
done
cat $out/ab.txt
for m in large; do echo "$m planar48 $(timeout 300 python bench.py --model $m --steps 15 --warmup 5 --no-extras 2>/dev/null | tail -1 | cut -c150-200)"; echo "$m rowmajor $(HSIMAE_WGRAD_PLANAR=0 timeout 300 python bench.py --model $m --steps 15 --warmup 5 --no-extras 2>/dev/null | tail -1 | cut -c150-200)"; done | tee $out/large.txt
timeout 900 python3 scripts/phase_timing.py > $out/phase_timing.txt 2>&1; grep -A 12 "dec_bwd_attn" $out/phase_timing.txt | head -40
cd /tmp && export TMPDIR=/tmp
for v in pad48 pad0 noswz; do
  lib=; [ $v != pad48 ] && lib="$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so"
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-extras > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_$v/*/*kernel_stats.csv | head -1); head -9 $f | cut -d, -f1-4 | sed "s/^/$v /" | cut -c1-200
done
