#!/bin/bash
# D = 256 swizzle retry (3 spills now) + blk256 LATE_X: Large A/B and kernel stats
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_q}; out=gpurun_out/$tag; mkdir -p $out
HSIMAE_LIB=$PWD/variants/swz256/libhsimae_hip.so timeout 600 python -m pytest tests -m gpu -x -q -k "large_n64 or wide_configs or d256" > $out/pytest_sel.txt 2>&1; tail -2 $out/pytest_sel.txt
for i in 1 2; do
  for v in default swz256; do
    lib=$PWD/hsimae_amd/libhsimae_hip.so; [ $v != default ] && lib=$PWD/variants/$v/libhsimae_hip.so
    HSIMAE_LIB=$lib timeout 300 python bench.py --model large --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | cut -c60-175 | sed "s/^/large $v /"
  done
done
cd /tmp && export TMPDIR=/tmp
for v in default swz256; do
  lib=$GRAFT_REPO_ROOT/hsimae_amd/libhsimae_hip.so; [ $v != default ] && lib=$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/st" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model large --steps 3 --warmup 2 --no-extras > /dev/null 2>&1
  echo "large $v: $(grep 'enc_mlp\|blk256' $GRAFT_REPO_ROOT/$out/st/*/*_kernel_stats.csv | sed 's/(anonymous namespace):://g' | cut -d, -f1,4 | cut -c6-60 | tr '\n' ' ')"
  rm -rf $GRAFT_REPO_ROOT/$out/st
done
