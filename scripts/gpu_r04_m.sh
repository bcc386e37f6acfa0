#!/bin/bash
# in-order vmcnt fixes: enc_mlp_bwd epilogue re-reads ahead of the last stores; decoder backward kernels spill-free with the
# next-sample prefetch behind the L2-hot re-reads.  Parity subset + Base A/B + kernel stats
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_m}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "mlp or c1_base48 or tiny or decoder or dec_block or padded" > $out/pytest_sel.txt 2>&1; tail -3 $out/pytest_sel.txt
for i in 1 2 3; do
  for v in pre_persist default; do
    lib=$PWD/hsimae_amd/libhsimae_hip.so; [ $v != default ] && lib=$PWD/variants/$v/libhsimae_hip.so
    HSIMAE_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 10 --no-extras 2>/dev/null | tail -1 | cut -c60-175 | sed "s/^/base $v /"
  done
done
cd /tmp && export TMPDIR=/tmp
for v in pre_persist default; do
  lib=$GRAFT_REPO_ROOT/hsimae_amd/libhsimae_hip.so; [ $v != default ] && lib=$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/st" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-extras > /dev/null 2>&1
  echo "base $v: $(grep 'enc_mlp\|dec_bwd' $GRAFT_REPO_ROOT/$out/st/*/*_kernel_stats.csv | sed 's/(anonymous namespace):://g' | cut -d, -f1,4 | cut -c6-60 | tr '\n' ' ')"
  rm -rf $GRAFT_REPO_ROOT/$out/st
done
