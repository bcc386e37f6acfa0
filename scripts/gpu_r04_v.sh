#!/bin/bash
# decoder MLP half forward (enc_mlp_fwd_kernel<64,192>): 64- vs 96- vs 128-row panels
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_v}; out=gpurun_out/$tag; mkdir -p $out
for r in 96 128; do
HSIMAE_LIB=$PWD/variants/rf$r/libhsimae_hip.so timeout 600 python -m pytest tests -m gpu -x -q -k "c2_full or config1 or tiny or decoder or fused_encoder_mlp" > $out/pytest_rf$r.txt 2>&1; tail -1 $out/pytest_rf$r.txt
done
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2 3; do
  echo "r64    $(b)" >> $out/ab.txt
  echo "r96    $(HSIMAE_LIB=$PWD/variants/rf96/libhsimae_hip.so b)" >> $out/ab.txt
  echo "r128   $(HSIMAE_LIB=$PWD/variants/rf128/libhsimae_hip.so b)" >> $out/ab.txt
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in 64 96 128; do
if [ $v = 64 ]; then unset HSIMAE_LIB; else export HSIMAE_LIB=$GRAFT_REPO_ROOT/variants/rf$v/libhsimae_hip.so; fi
HSIMAE_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | grep -c metric
python3 - "$GRAFT_REPO_ROOT/$out/stats" $v <<'PY' | tee -a $GRAFT_REPO_ROOT/$out/variants.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'enc_mlp_fwd_kernel<64' in r['Name']: print('R =',sys.argv[2], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
rm -rf $GRAFT_REPO_ROOT/$out/stats
done
