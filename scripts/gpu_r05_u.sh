#!/bin/bash
# round 5, final build: launch-shape knob sweep (as scripts/gpu_knob_sweep_r04.sh) + PMC traffic files on the final sources
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_u; mkdir -p $out
for m in base large; do bash scripts/gpu_step_traffic.sh $m > /dev/null 2>&1; cp gpurun_out/step_traffic_$m.json $out/; done
bash scripts/gpu_step_traffic.sh huge > /dev/null 2>&1; cp gpurun_out/step_traffic_huge.json $out/step_traffic_huge_fp8.json
rm -rf gpurun_out/traffic_*
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
b > /dev/null
for rep in 1 2; do
echo "default                      $(b)" | tee -a $out/knob_sweep.txt
for kv in HSIMAE_WGRAD_WGS=384 HSIMAE_WGRAD_WGS=640 HSIMAE_WGRAD_WGS=768 HSIMAE_WGRAD_DS=4 HSIMAE_BLK128_WGS=512 HSIMAE_BLK128_BWD_WGS=512 HSIMAE_BLK128_BWD_WGS=192 HSIMAE_DEC_FWD_WGS=512 GPU_MAX_HW_QUEUES=2 GPU_MAX_HW_QUEUES=4 HSIMAE_TWO_STREAMS=0 HSIMAE_WGRAD_PLANAR=0; do
  echo "$kv   $(export $kv; b)" | tee -a $out/knob_sweep.txt
done
done
