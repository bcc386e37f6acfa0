#!/bin/bash
# round 5: start stagger of the persistent decoder backward kernels (classes of workgroups a fraction of a sample apart); MX-operand oracle test
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_g; mkdir -p $out
b() { timeout 300 python bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
for rep in 1 2 3; do
  echo "base  $(b)" >> $out/ab.txt
  for v in stg2 stg4 stg4b stg8; do echo "$v $(HSIMAE_LIB=variants/$v/libhsimae_hip.so b)" >> $out/ab.txt; done
done
cat $out/ab.txt
timeout 900 python -m pytest tests/test_gpu_fp8.py -m gpu -x -q -s -k "mx_operand" > $out/pytest_mx.txt 2>&1; grep -E "^\[mx|passed|failed|assert" $out/pytest_mx.txt | cut -c1-600
cd /tmp && export TMPDIR=/tmp
for v in base stg2 stg4 stg8; do
  lib=; [ $v != base ] && lib="$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so"
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-extras > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_$v/*/*kernel_stats.csv | head -1); grep -E "dec_bwd|dec_attn_fwd" $f | cut -d, -f1-4 | sed "s/^/$v /" | cut -c1-200
done
