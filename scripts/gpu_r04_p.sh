#!/bin/bash
# pair launches of the two axis stacks: full GPU suite + Base A/B (HSIMAE_PAIR_LAUNCH=0 = the two-stream schedule) + kernel stats
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_p}; out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt
for i in 1 2 3; do
  HSIMAE_PAIR_LAUNCH=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-extras 2>/dev/null | tail -1 | cut -c60-175 | sed "s/^/two-stream /"
  timeout 300 python bench.py --steps 40 --warmup 10 --no-extras 2>/dev/null | tail -1 | cut -c60-175 | sed "s/^/pair       /"
done
HSIMAE_PAIR_LAUNCH=0 HSIMAE_TWO_STREAMS=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-extras 2>/dev/null | tail -1 | cut -c60-175 | sed "s/^/one stream /"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/st" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-extras > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; cp $out/st/*/*_kernel_stats.csv $out/kernel_stats_base_pair.csv; rm -rf $out/st
head -12 $out/kernel_stats_base_pair.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-110
