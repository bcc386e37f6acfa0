#!/bin/bash
# SQ / MFMA counters of the base step + a stall-attribution pass
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_c}; out=gpurun_out/$tag; mkdir -p $out
bash scripts/gpu_counters_model.sh base > /dev/null 2>&1
cp gpurun_out/counters_base_sq.txt $out/sq_counters_base.txt; cp gpurun_out/counters_base_mfma.txt $out/mfma_counters_base.txt
cd /tmp && export TMPDIR=/tmp; export HSIMAE_TWO_STREAMS=0
o2="$GRAFT_REPO_ROOT/gpurun_out/counters_stall"; rm -rf $o2; mkdir -p $o2
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $o2 -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-extras 2>&1 | grep -c metric
cd "$GRAFT_REPO_ROOT"; python scripts/pmc_summary.py $o2 > $out/stall_counters_base.txt
cut -c1-200 $out/sq_counters_base.txt | head -14; cut -c1-220 $out/stall_counters_base.txt | head -14
