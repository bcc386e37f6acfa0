#!/bin/bash
# round 5: timing ablation of the L2 -> CU weight stream of the row-panel MLP kernels (-DHS_ABL_WSTREAM: every hidden chunk computes
# with chunk 0's fragments; results are wrong on purpose) = the ceiling of any scheme that shares weight fragments between panels
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_o; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in base large; do
for v in base abl_wstream; do
  lib=; [ $v != base ] && lib="$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so"
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_${m}_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $m --steps 6 --warmup 2 --no-extras --no-verify > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_${m}_$v/*/*kernel_stats.csv | head -1); grep -E "enc_mlp" $f | cut -d, -f1-4 | sed "s/^/$m $v /" | sed 's/(anonymous namespace):://g' | cut -c1-140
done; done | tee $GRAFT_REPO_ROOT/$out/summary.txt
