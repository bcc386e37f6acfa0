#!/bin/bash
# SQ + MFMA counters per kernel for one model (single stream): bash scripts/gpu_counters_model.sh <model> [bench args]
model=$1; shift
cd /tmp && export TMPDIR=/tmp
export HSIMAE_TWO_STREAMS=0
out="$GRAFT_REPO_ROOT/gpurun_out/counters_$model"; rm -rf "$out"; mkdir -p "$out/sq" "$out/mfma"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$out/sq" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $model --steps 1 --warmup 1 --no-extras "$@" 2>&1 | grep -c metric
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/mfma" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $model --steps 1 --warmup 1 --no-extras "$@" 2>&1 | grep -c metric
cd "$GRAFT_REPO_ROOT"
python scripts/pmc_summary.py $out/sq > gpurun_out/counters_${model}_sq.txt
python scripts/pmc_summary.py $out/mfma > gpurun_out/counters_${model}_mfma.txt
head -14 gpurun_out/counters_${model}_mfma.txt | cut -c1-170
