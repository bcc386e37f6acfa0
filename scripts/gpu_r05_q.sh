#!/bin/bash
# round 5: the weights-resident persistent MLP forward kernel at D = 128 (HSIMAE_MLP_FWD_RES=1) against the panel kernel
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_q; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_encoder_mlp_half" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
b() { timeout 300 python bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
b > /dev/null
for rep in 1 2 3; do
  echo "panel  $(HSIMAE_MLP_FWD_RES=0 b)" >> $out/ab.txt
  echo "res256 $(HSIMAE_MLP_FWD_RES=1 b)" >> $out/ab.txt
  echo "res512 $(HSIMAE_MLP_FWD_RES=1 HSIMAE_MLP_FWD_RES_WGS=512 b)" >> $out/ab.txt
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  HSIMAE_MLP_FWD_RES=$v HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-extras --no-verify > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_$v/*/*kernel_stats.csv | head -1); grep -E "enc_mlp_fwd|blk128_fwd" $f | cut -d, -f1-4 | sed "s/^/res=$v /" | sed 's/(anonymous namespace):://g' | cut -c1-140
done
