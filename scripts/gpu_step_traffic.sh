#!/bin/bash
# usage (GPU box): gpu_step_traffic.sh <model> [bench args]: FETCH_SIZE / WRITE_SIZE passes of a 2-step bench run -> gpurun_out/step_traffic_<model>.json
# (single stream, so that the kernel names match the single-stream kernel stats that scripts/kernel_table.py joins them with)
export HSIMAE_TWO_STREAMS=0
model=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$GRAFT_REPO_ROOT/gpurun_out/traffic_${model}_$c"; mkdir -p "$GRAFT_REPO_ROOT/gpurun_out/traffic_${model}_$c"
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/traffic_${model}_$c" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $model --steps 1 --warmup 1 --no-extras "$@" 2>&1 | grep -c metric
done
cd "$GRAFT_REPO_ROOT"
python scripts/step_traffic.py gpurun_out/traffic_${model}_FETCH_SIZE gpurun_out/traffic_${model}_WRITE_SIZE 2 gpurun_out/step_traffic_${model}.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bench.py --model $model --steps 1 --warmup 1 $*"
