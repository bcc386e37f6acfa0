#!/bin/bash
# full GPU suite on the default schedule + the axis-stack tests on the pair schedule
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_s}; out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
HSIMAE_PAIR_LAUNCH=1 timeout 900 python -m pytest tests -m gpu -x -q -k "c1_base48 or tiny or summary or config1 or c2_full or ddp or fused_attention_half_matches or fused_encoder_mlp" > $out/pytest_pair.txt 2>&1; tail -2 $out/pytest_pair.txt
