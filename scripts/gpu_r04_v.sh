#!/bin/bash
# blk128_fwd_kernel with four-wave workgroups (two heads per wave, two independent workgroups per CU): HSIMAE_BLK128_HPW=2
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_v}; out=gpurun_out/$tag; mkdir -p $out
HSIMAE_BLK128_HPW=2 timeout 900 python -m pytest tests -m gpu -x -q -k "fused_attention_half or c2_full or config1 or tiny or c1_base48 or droppath or dualvit" > $out/pytest_hpw2.txt 2>&1; tail -2 $out/pytest_hpw2.txt
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2 3; do
  echo "hpw1   $(b)" >> $out/ab.txt
  echo "hpw2   $(HSIMAE_BLK128_HPW=2 b)" >> $out/ab.txt
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in 1 2; do
HSIMAE_BLK128_HPW=$v HSIMAE_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | grep -c metric
echo "hpw$v: $(grep 'blk128_fwd' $GRAFT_REPO_ROOT/$out/stats/*/*_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140)" | tee -a $GRAFT_REPO_ROOT/$out/variants.txt
rm -rf $GRAFT_REPO_ROOT/$out/stats
done
