#!/bin/bash
# round 5: dec_bwd_mlp with the next sample's x1 / dY rows arriving by LDS-DMA (default build) against the fetch at the top of the sample (variant nodma)
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_r; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -m gpu -x -q -k "decoder or c1_base48 or tiny or dec_block" > $out/pytest_dec.txt 2>&1; tail -3 $out/pytest_dec.txt
b() { timeout 300 python bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
b > /dev/null
for rep in 1 2 3; do
  echo "dma   $(b)" >> $out/ab.txt
  echo "nodma $(HSIMAE_LIB=variants/nodma/libhsimae_hip.so b)" >> $out/ab.txt
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in base nodma; do
  lib=; [ $v != base ] && lib="$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so"
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-extras --no-verify > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_$v/*/*kernel_stats.csv | head -1); grep -E "dec_bwd" $f | cut -d, -f1-4 | sed "s/^/$v /" | sed 's/(anonymous namespace):://g' | cut -c1-120
done
