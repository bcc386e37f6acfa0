#!/bin/bash
# round 4: decoder backward LDS layouts — parity tests, same-box A/B against the round-3 library, kernel stats
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_b}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "decoder or dec_block or padded_key or fused" > $out/pytest_dec.txt 2>&1; tail -3 $out/pytest_dec.txt
for i in 1 2; do
  HSIMAE_LIB=$PWD/variants/r03/libhsimae_hip.so timeout 300 python bench.py --steps 40 --warmup 10 --no-extras 2>/dev/null | tail -1 | cut -c1-200 | tee -a $out/ab_old.txt
  timeout 300 python bench.py --steps 40 --warmup 10 --no-extras 2>/dev/null | tail -1 | cut -c1-200 | tee -a $out/ab_new.txt
done
cd /tmp && export TMPDIR=/tmp
HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_new" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-extras > /dev/null 2>&1
HSIMAE_LIB=$GRAFT_REPO_ROOT/variants/r03/libhsimae_hip.so HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_old" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-extras > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
for v in new old; do cp $out/stats_$v/*/*_kernel_stats.csv $out/kernel_stats_$v.csv; rm -rf $out/stats_$v; echo $v; head -12 $out/kernel_stats_$v.csv | cut -d, -f1-4 | cut -c1-100; done
