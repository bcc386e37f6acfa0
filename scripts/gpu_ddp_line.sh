#!/bin/bash
# the 1-rank data-parallel bench line next to the plain one, same box
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/ddp_line; mkdir -p $out
timeout 300 python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $out/plain.json; cut -c1-200 $out/plain.json
timeout 300 python bench.py --gpus 1 --force-ddp --steps 50 --warmup 10 --no-extras 2>/dev/null | tail -1 > $out/ddp.json; cut -c1-200 $out/ddp.json
HSIMAE_KEEP_HW_QUEUES=1 timeout 300 python bench.py --gpus 1 --force-ddp --steps 50 --warmup 10 --no-extras 2>/dev/null | tail -1 > $out/ddp_default_queues.json; cut -c1-200 $out/ddp_default_queues.json
grep -o '"comm": {[^}]*}' $out/ddp.json $out/ddp_default_queues.json
