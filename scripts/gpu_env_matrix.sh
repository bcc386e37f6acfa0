#!/bin/bash
# end-to-end parity tests under the A/B switches that change which kernels run (fallback schedules must keep working)
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/env_matrix.txt; : > $out
K="c2_full or config1 or tiny or c1_base48 or wide_configs or large_n64 or encode_backward or split_api"
for e in "HSIMAE_TWO_STREAMS=0" "HSIMAE_FUSED_ATTN_BLOCK=0" "HSIMAE_FUSED_ATTN_BLOCK_BWD=0" "HSIMAE_ATTN_BWD_RECOMPUTE=0" "HSIMAE_FUSED_MLP=0" "HSIMAE_FUSED_PROJ_BWD=0" "HSIMAE_FUSED_LNBWD=0" "HSIMAE_WGRAD_PLANAR=0" "HSIMAE_FUSED_ATTN_BLOCK256=0" "HSIMAE_FUSED_ATTN_BLOCK256_BWD=0" "HSIMAE_DEC_SPLIT=0" "HSIMAE_WGRAD_SLAB=0" "HSIMAE_DETERMINISTIC=1" "HSIMAE_FUSED_DEC=0" "HSIMAE_WGRAD_DMA=0"; do
  r=$(env $e timeout 900 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -1)
  echo "$e: $r" | tee -a $out
done
