#!/bin/bash
# round-3 A/B of the split decoder forward (GPU box): bash scripts/gpu_r03_a.sh <tag>
tag=${1:-r03_b}
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -k "fused_decoder or tiny_and_odd or c2_full" 2>&1 | tail -5
for sp in 0 1; do
  HSIMAE_DEC_SPLIT=$sp timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $out/bench_split$sp.json; cut -c1-200 $out/bench_split$sp.json
done
cd /tmp && export TMPDIR=/tmp
HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_single" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-extras 2>&1 | grep -c metric
cd "$GRAFT_REPO_ROOT"; cp $out/stats_single/*/*_kernel_stats.csv $out/kernel_stats_single.csv; head -16 $out/kernel_stats_single.csv | cut -c1-150
