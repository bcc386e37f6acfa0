#!/bin/bash
# round 5: what enc_mlp_fwd would take without its memory waits (timing ablations, wrong results on purpose): no row fetch, + no weight
# stream, + no stores = the compute floor of the 3-workgroups-per-CU form
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in base large; do
for v in base abl_wstream abl_fwd_noload abl_fwd_noload_ws abl_fwd_nomem; do
  lib=; [ $v != base ] && lib="$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so"
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_${m}_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $m --steps 6 --warmup 2 --no-extras --no-verify > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_${m}_$v/*/*kernel_stats.csv | head -1); grep -E "enc_mlp_fwd" $f | cut -d, -f1-4 | sed "s/^/$m $v /" | sed 's/(anonymous namespace):://g' | cut -c1-140
done; done | tee $GRAFT_REPO_ROOT/$out/summary.txt
