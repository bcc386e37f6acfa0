#!/bin/bash
# encoder MLP kernels with swizzled panels: parity subset, Base / Large A/B against the build before it, kernel stats
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_j}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "mlp or fused_enc or c1_base48 or large_n64 or dualvit or widths" > $out/pytest_sel.txt 2>&1; tail -3 $out/pytest_sel.txt
for i in 1 2; do
  for v in pre_swz default; do
    lib=$PWD/hsimae_amd/libhsimae_hip.so; [ $v != default ] && lib=$PWD/variants/$v/libhsimae_hip.so
    HSIMAE_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 10 --no-extras 2>/dev/null | tail -1 | cut -c60-175 | sed "s/^/base $v /"
    HSIMAE_LIB=$lib timeout 300 python bench.py --model large --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | cut -c60-175 | sed "s/^/large $v /"
  done
done
cd /tmp && export TMPDIR=/tmp
for m in base large; do for v in pre_swz default; do
  lib=$GRAFT_REPO_ROOT/hsimae_amd/libhsimae_hip.so; [ $v != default ] && lib=$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/st" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $m --steps 3 --warmup 2 --no-extras > /dev/null 2>&1
  echo "$m $v: $(grep enc_mlp $GRAFT_REPO_ROOT/$out/st/*/*_kernel_stats.csv | sed 's/(anonymous namespace):://g' | cut -d, -f1,4 | cut -c6-60 | tr '\n' ' ')"
  rm -rf $GRAFT_REPO_ROOT/$out/st
done; done
