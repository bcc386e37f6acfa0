#!/bin/bash
# round 5: wgrad_dma does not fetch the 16-byte pieces past N / K of edge tiles (default build) against fetching them (variant noedge)
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_s; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -x -q -k "wgrad or weight_gradient or c1_base48 or tiny or c2_full or large_n64 or wide" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
b() { timeout 300 python bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
b > /dev/null
for rep in 1 2 3; do
  echo "edge-skip $(b)" >> $out/ab.txt
  echo "noedge    $(HSIMAE_LIB=variants/noedge/libhsimae_hip.so b)" >> $out/ab.txt
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for m in base large; do
for v in base noedge; do
  lib=; [ $v != base ] && lib="$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so"
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_${m}_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $m --steps 10 --warmup 3 --no-extras --no-verify > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_${m}_$v/*/*kernel_stats.csv | head -1); grep -E "wgrad" $f | cut -d, -f1-4 | sed "s/^/$m $v /" | sed 's/(anonymous namespace):://g' | cut -c1-120
done; done
