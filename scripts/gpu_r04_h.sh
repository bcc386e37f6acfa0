#!/bin/bash
# SQ / stall counters of the Large step (single stream)
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_h}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; export HSIMAE_TWO_STREAMS=0
o1="$GRAFT_REPO_ROOT/gpurun_out/c_large_sq"; o2="$GRAFT_REPO_ROOT/gpurun_out/c_large_stall"; rm -rf $o1 $o2; mkdir -p $o1 $o2
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $o1 -- python3 "$GRAFT_REPO_ROOT/bench.py" --model large --steps 1 --warmup 1 --no-extras 2>&1 | grep -c metric
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $o2 -- python3 "$GRAFT_REPO_ROOT/bench.py" --model large --steps 1 --warmup 1 --no-extras 2>&1 | grep -c metric
cd "$GRAFT_REPO_ROOT"; python scripts/pmc_summary.py $o1 > $out/sq_counters_large.txt; python scripts/pmc_summary.py $o2 > $out/stall_counters_large.txt
cut -c1-230 $out/sq_counters_large.txt | head -8; cut -c1-230 $out/stall_counters_large.txt | head -8
