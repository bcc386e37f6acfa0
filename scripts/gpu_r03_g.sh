#!/bin/bash
tag=${1:-r03_g}
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -5
for cp in 0 1; do
  HSIMAE_MLP_COPIER=$cp timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $out/bench_copier$cp.json; cut -c1-200 $out/bench_copier$cp.json
done
cd /tmp && export TMPDIR=/tmp
HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_single" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-extras 2>&1 | grep -c metric
cd "$GRAFT_REPO_ROOT"; cp $out/stats_single/*/*_kernel_stats.csv $out/kernel_stats_single.csv; head -14 $out/kernel_stats_single.csv | cut -c1-150
