#!/bin/bash
# blk256_fwd variants: kernel time per launch from a short single-stream profile of bench.py --model large
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_g}; out=gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q -k "d256" > $out/pytest_sel.txt 2>&1; tail -2 $out/pytest_sel.txt
cd /tmp && export TMPDIR=/tmp
for v in default w256_early w256_pf2; do
  lib=$GRAFT_REPO_ROOT/hsimae_amd/libhsimae_hip.so; [ $v != default ] && lib=$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/st_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model large --steps 3 --warmup 2 --no-extras > /dev/null 2>&1
  echo "$v: $(grep blk256 $GRAFT_REPO_ROOT/$out/st_$v/*/*_kernel_stats.csv | cut -d, -f2-4)"
  cp $GRAFT_REPO_ROOT/$out/st_$v/*/*_kernel_stats.csv $GRAFT_REPO_ROOT/$out/kernel_stats_large_$v.csv; rm -rf $GRAFT_REPO_ROOT/$out/st_$v
done
cd "$GRAFT_REPO_ROOT"
for v in default w256_early; do
  lib=$PWD/hsimae_amd/libhsimae_hip.so; [ $v != default ] && lib=$PWD/variants/$v/libhsimae_hip.so
  HSIMAE_LIB=$lib timeout 300 python bench.py --model large --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | cut -c60-200 | sed "s/^/$v /"
done
