#!/bin/bash
# round 5: order of the attention core of dec_bwd_attn (HS_DEC_CORE_ORDER: 1 = dk/dv of the previous tile pinned behind this tile's
# softmax arithmetic, 4 = score / dP products one tile ahead, 5 = both); + the new bench self-check test
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_n; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_bench.py -m gpu -x -q > $out/pytest_bench.txt 2>&1; tail -2 $out/pytest_bench.txt
b() { timeout 300 python bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
b > /dev/null
for rep in 1 2 3; do
  echo "base  $(b)" >> $out/ab.txt
  for v in co1 co4 co5; do echo "$v $(HSIMAE_LIB=variants/$v/libhsimae_hip.so b)" >> $out/ab.txt; done
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in base co1 co4 co5; do
  lib=; [ $v != base ] && lib="$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so"
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-extras > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_$v/*/*kernel_stats.csv | head -1); grep -E "dec_bwd" $f | cut -d, -f1-4 | sed "s/^/$v /" | sed 's/(anonymous namespace):://g' | cut -c1-120
done
for v in co5; do HSIMAE_LIB=$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so timeout 600 python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_e2e.py $GRAFT_REPO_ROOT/tests/test_gpu_kernels.py -m gpu -x -q -k "decoder or c1_base48 or tiny or dec_block" 2>&1 | tail -2; done
