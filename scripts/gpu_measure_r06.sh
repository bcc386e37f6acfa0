#!/bin/bash
# round-6 measurement set (GPU box): bash scripts/gpu_measure_r06.sh <tag>
#   per-kernel HBM traffic (PMC) first, so that the bench lines carry THIS build's counters in roofline.traffic;
#   bench lines (base with extras + CPU baseline; large; huge fp8 / bf16; base through the 1-rank RCCL path);
#   single- and two-stream kernel stats; SQ / MFMA / stall counters of the base and large steps
tag=${1:-r06_end}
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/$tag; mkdir -p $out
for m in base large; do bash scripts/gpu_step_traffic.sh $m > /dev/null 2>&1; cp gpurun_out/step_traffic_$m.json $out/; cp gpurun_out/step_traffic_$m.json profiles/; done
bash scripts/gpu_step_traffic.sh huge > /dev/null 2>&1; cp gpurun_out/step_traffic_huge.json $out/step_traffic_huge_fp8.json; cp gpurun_out/step_traffic_huge.json profiles/step_traffic_huge_fp8.json
timeout 600 python bench.py --steps 20 --warmup 5 2> $out/bench_base.err | tail -1 > $out/bench_base.json; cut -c1-220 $out/bench_base.json
timeout 300 python bench.py --model large --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_large.json; cut -c1-220 $out/bench_large.json
timeout 300 python bench.py --model huge --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_huge_fp8.json; cut -c1-220 $out/bench_huge_fp8.json
timeout 300 python bench.py --model huge --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_huge_bf16.json; cut -c1-220 $out/bench_huge_bf16.json
timeout 300 python bench.py --gpus 1 --force-ddp --steps 50 --warmup 10 --no-extras 2>/dev/null | tail -1 > $out/bench_base_ddp_path_1rank.json; cut -c1-220 $out/bench_base_ddp_path_1rank.json
cd /tmp && export TMPDIR=/tmp
for m in base large huge; do
  HSIMAE_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_${m}_single" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $m --steps 3 --warmup 2 --no-extras 2>&1 | grep -c metric
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_base_two" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-extras 2>&1 | grep -c metric
cd "$GRAFT_REPO_ROOT"; for m in base large huge; do cp $out/stats_${m}_single/*/*_kernel_stats.csv $out/kernel_stats_${m}_single_stream.csv; rm -rf $out/stats_${m}_single; done
cp $out/stats_base_two/*/*_kernel_stats.csv $out/kernel_stats_base_two_streams.csv; rm -rf $out/stats_base_two
for m in base large; do
  bash scripts/gpu_counters_model.sh $m > /dev/null 2>&1; cp gpurun_out/counters_${m}_sq.txt $out/sq_counters_$m.txt; cp gpurun_out/counters_${m}_mfma.txt $out/mfma_counters_$m.txt
done
cd /tmp; export HSIMAE_TWO_STREAMS=0
o2="$GRAFT_REPO_ROOT/gpurun_out/counters_stall"; rm -rf $o2; mkdir -p $o2
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $o2 -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-extras 2>&1 | grep -c metric
cd "$GRAFT_REPO_ROOT"; python scripts/pmc_summary.py $o2 > $out/stall_counters_base.txt
rm -rf gpurun_out/traffic_* gpurun_out/counters_base gpurun_out/counters_large gpurun_out/counters_stall
ls $out | head -40
