#!/bin/bash
# round 5: phase profile of the decoder backward kernels + timing ablations (upper bounds of three candidate changes), same box
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_c; mkdir -p $out
b() { timeout 300 python bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
for rep in 1 2; do
  echo "base            $(b)" >> $out/ab.txt
  for v in abl_dw2 abl_dec_reread abl_planar; do echo "$v $(HSIMAE_LIB=variants/$v/libhsimae_hip.so b)" >> $out/ab.txt; done
done
cat $out/ab.txt
timeout 900 python3 scripts/phase_timing.py > $out/phase_timing.txt 2>&1; tail -40 $out/phase_timing.txt
cd /tmp && export TMPDIR=/tmp
for v in base abl_dw2 abl_planar abl_dec_reread; do
  lib=; [ $v != base ] && lib="$GRAFT_REPO_ROOT/variants/$v/libhsimae_hip.so"
  HSIMAE_LIB=$lib HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats_$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-extras > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/stats_$v/*/*kernel_stats.csv | head -1); head -12 $f | cut -d, -f1-4 | sed "s/^/$v /"
done
