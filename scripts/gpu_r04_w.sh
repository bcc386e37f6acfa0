#!/bin/bash
# attn16_bwd_kernel: delta from P * dP inside the core for NT <= 4 (default) vs the prologue form (variants/attn16_old): Large / Huge
cd "$GRAFT_REPO_ROOT"; tag=${1:-r04_w}; out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q -k "large or huge or wide or fp8 or attn or widths or fused_attention_half_d256 or c2_full or config1" > $out/pytest_sel.txt 2>&1; tail -2 $out/pytest_sel.txt
b() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2; do
  echo "large old  $(HSIMAE_LIB=$PWD/variants/attn16_old/libhsimae_hip.so b --model large)" >> $out/ab.txt
  echo "large new  $(b --model large)" >> $out/ab.txt
  echo "huge  old  $(HSIMAE_LIB=$PWD/variants/attn16_old/libhsimae_hip.so b --model huge)" >> $out/ab.txt
  echo "huge  new  $(b --model huge)" >> $out/ab.txt
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in attn16_old default; do
if [ $v = attn16_old ]; then export HSIMAE_LIB=$GRAFT_REPO_ROOT/variants/attn16_old/libhsimae_hip.so; else unset HSIMAE_LIB; fi
HSIMAE_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --model large --steps 3 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | grep -c metric
python3 - "$GRAFT_REPO_ROOT/$out/stats" $v <<'PY' | tee -a $GRAFT_REPO_ROOT/$out/variants.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'attn16_bwd' in r['Name']: print(sys.argv[2], r['Name'][35:75], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
rm -rf $GRAFT_REPO_ROOT/$out/stats
done
