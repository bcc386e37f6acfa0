#!/bin/bash
# does the pair-launch argument indexing (pp.v[blockIdx.y]) cost the single launches anything?  kernel time per step, same box
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_r; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in base large; do for v in default nopair swz256 nopair_swz256; do
  lib=$GRAFT_REPO_ROOT/hsimae_amd/libhsimae_hip.so; [ $v != default ] && lib=$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/st" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $m --steps 3 --warmup 2 --no-extras > /dev/null 2>&1
  echo "$m $v: $(grep 'enc_mlp\|lnbwd_dma' $GRAFT_REPO_ROOT/$out/st/*/*_kernel_stats.csv | sed 's/(anonymous namespace):://g; s/"//g' | awk -F, '{printf "%s=%.1fus ", substr($1,6,28), $(NF-4)/1000}')"
  rm -rf $GRAFT_REPO_ROOT/$out/st
done; done
