#!/bin/bash
# round 5: full GPU suite on the current build + the round's kernel changes as one same-box A/B (r04-equivalent build: no stagger, 272-byte
# pitch in blk128_fwd, row-major operands via HSIMAE_WGRAD_PLANAR=0)
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_j; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $out/pytest_gpu.txt 2>&1; tail -14 $out/pytest_gpu.txt
b() { timeout 300 python bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
b > /dev/null
for rep in 1 2 3 4; do
  echo "round-5 build            $(b)" >> $out/ab.txt
  echo "round-4 equivalent build $(HSIMAE_WGRAD_PLANAR=0 HSIMAE_LIB=variants/r04eq/libhsimae_hip.so b)" >> $out/ab.txt
done
cat $out/ab.txt
