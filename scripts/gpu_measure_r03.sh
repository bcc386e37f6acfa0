#!/bin/bash
# round-3 measurement set (GPU box): bash scripts/gpu_measure_r03.sh <tag>
#   bench lines (base with extras + CPU baseline; large; huge fp8 / bf16; base through the 1-rank RCCL path), single- and two-stream
#   kernel stats, per-kernel HBM traffic (PMC), SQ / MFMA counters of the base step
tag=${1:-r03_f}
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/$tag; mkdir -p $out
# traffic first, so that the bench lines below carry THIS run's counters in roofline.traffic (bench.py reads profiles/step_traffic_*.json)
for m in base large; do bash scripts/gpu_step_traffic.sh $m > /dev/null 2>&1; cp gpurun_out/step_traffic_$m.json $out/; cp gpurun_out/step_traffic_$m.json profiles/; done
bash scripts/gpu_step_traffic.sh huge > /dev/null 2>&1; cp gpurun_out/step_traffic_huge.json $out/step_traffic_huge_fp8.json; cp gpurun_out/step_traffic_huge.json profiles/step_traffic_huge_fp8.json
timeout 600 python bench.py 2> $out/bench_base.err | tail -1 > $out/bench_base.json; cut -c1-220 $out/bench_base.json
timeout 300 python bench.py --model large --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_large.json; cut -c1-220 $out/bench_large.json
timeout 300 python bench.py --model huge --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_huge_fp8.json; cut -c1-220 $out/bench_huge_fp8.json
timeout 300 python bench.py --model huge --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_huge_bf16.json; cut -c1-220 $out/bench_huge_bf16.json
timeout 300 python bench.py --gpus 1 --force-ddp --steps 50 --warmup 10 --no-extras 2>/dev/null | tail -1 > $out/bench_base_ddp_path_1rank.json; cut -c1-220 $out/bench_base_ddp_path_1rank.json
cd /tmp && export TMPDIR=/tmp
for m in base large huge; do
  HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_${m}_single" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $m --steps 3 --warmup 2 --no-extras 2>&1 | grep -c metric
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_base_two" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-extras 2>&1 | grep -c metric
cd "$GRAFT_REPO_ROOT"; for m in base large huge; do cp $out/stats_${m}_single/*/*_kernel_stats.csv $out/kernel_stats_${m}_single_stream.csv; done; cp $out/stats_base_two/*/*_kernel_stats.csv $out/kernel_stats_base_two_streams.csv
bash scripts/gpu_counters_model.sh base > /dev/null 2>&1; cp gpurun_out/counters_base_sq.txt $out/sq_counters_base.txt; cp gpurun_out/counters_base_mfma.txt $out/mfma_counters_base.txt
ls $out | head -40
