#!/bin/bash
# round 6: the round's kernel changes as ONE same-box A/B — the final library against the round-5 kernels behind the same ABI
# (variants/r5_kernels: the tree at b70b8b9, i.e. round 5's csrc + hsimae_build_info), four alternating repetitions of the two-stream step,
# plus the decoder-only and encoder-only HIP-event timings of each once
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; out=gpurun_out/r06_round_ab; mkdir -p $out
for rep in 1 2 3 4; do for L in variants/r5_kernels/libhsimae_hip.so hsimae_amd/libhsimae_hip.so; do
  echo "== $L" | tee -a $out/ab.txt
  HSIMAE_LIB="$R/$L" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('    ms_per_step', d['ms_per_step'], ' HIP-event median', d['step_ms']['median'])" | tee -a $out/ab.txt
done; done
for L in variants/r5_kernels/libhsimae_hip.so hsimae_amd/libhsimae_hip.so; do
  echo "== extras $L" | tee -a $out/ab.txt
  HSIMAE_LIB="$R/$L" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline_decoder',{}); e=d.get('encoder_mfma_frac',{})
print('    ms_per_step', d['ms_per_step'], ' decoder fwd / bwd ms', r.get('fwd_ms'), r.get('bwd_ms'), ' encoder ms', e.get('ms'), ' build', d['build']['kernel_source_hash'], d['build']['matches_sources'])" | tee -a $out/ab.txt
done
