#!/bin/bash
# per-kernel HBM traffic of one step for the three models on the current sources (refreshes profiles/step_traffic_*.json), then the
# end-to-end parity tests under every A/B switch
cd "$GRAFT_REPO_ROOT"
for m in base large; do bash scripts/gpu_step_traffic.sh $m > /dev/null 2>&1; cp gpurun_out/step_traffic_$m.json gpurun_out/final_step_traffic_$m.json; done
bash scripts/gpu_step_traffic.sh huge > /dev/null 2>&1; cp gpurun_out/step_traffic_huge.json gpurun_out/final_step_traffic_huge_fp8.json
python - <<'PY'
import json
for m in ("base","large","huge_fp8"):
    d=json.load(open(f"gpurun_out/final_step_traffic_{m}.json")); print(m, d.get("kernel_source_sha"), d.get("total_bytes") or d.get("bytes_per_step"))
PY
bash scripts/gpu_env_matrix.sh
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
