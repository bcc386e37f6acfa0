#!/bin/bash
# whole library compiled for gfx950:xnack- (the pool runs with XNACK off) vs the default target (xnack "any")
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_v}; out=gpurun_out/$tag; mkdir -p $out
HSIMAE_LIB=$PWD/variants/xnackoff/libhsimae_hip.so timeout 600 python -m pytest tests -m gpu -x -q -k "c2_full or config1 or tiny or fused_attention_half_backward" > $out/pytest_x.txt 2>&1; tail -1 $out/pytest_x.txt
b() { timeout 300 python bench.py "$@" --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2 3; do
  echo "base default  $(b)" >> $out/ab.txt
  echo "base xnack-   $(HSIMAE_LIB=$PWD/variants/xnackoff/libhsimae_hip.so b)" >> $out/ab.txt
done
for i in 1 2; do
  echo "large default $(b --model large)" >> $out/ab.txt
  echo "large xnack-  $(HSIMAE_LIB=$PWD/variants/xnackoff/libhsimae_hip.so b --model large)" >> $out/ab.txt
done
cat $out/ab.txt
