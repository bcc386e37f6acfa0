#!/bin/bash
# rectangular-tile weight gradients: loop time without the commits (variants/nocommit), both kernels
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_x}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export HSIMAE_LIB=$GRAFT_REPO_ROOT/variants/nocommit/libhsimae_hip.so
for v in "base" "HSIMAE_WGRAD_RECT=0"; do
  if [ "$v" != base ]; then export $v; fi
  HSIMAE_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | grep -c metric
  echo "nocommit $v: $(grep 'wgrad_rect\|wgrad_dma' $GRAFT_REPO_ROOT/$out/stats/*/*_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160)" | tee -a $GRAFT_REPO_ROOT/$out/variants.txt
  rm -rf $GRAFT_REPO_ROOT/$out/stats
done
