#!/bin/bash
# launch-shape knobs on the final build, same box, bench.py --steps 40 --warmup 10 --no-extras (ms per step)
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/knob_sweep_r04.txt; : > $out
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for rep in 1 2; do
echo "default                      $(b)" | tee -a $out
for kv in HSIMAE_WGRAD_WGS=640 HSIMAE_WGRAD_WGS=896 HSIMAE_WGRAD_DS=4 HSIMAE_BLK128_WGS=512 HSIMAE_BLK128_BWD_WGS=512 HSIMAE_BLK128_BWD_WGS=192 HSIMAE_DEC_FWD_WGS=512 HSIMAE_MLP_FWD_WGS=1024 GPU_MAX_HW_QUEUES=2 GPU_MAX_HW_QUEUES=4 HSIMAE_TWO_STREAMS=0; do
  echo "$kv   $(export $kv; b)" | tee -a $out
done
done
