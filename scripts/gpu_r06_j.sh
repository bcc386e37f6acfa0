#!/bin/bash
# round 6, call J: the L2-exchange micro-benchmark again with the panel of ONE 48-row enc_mlp_bwd panel per CU (128 KB: 4-MB halves, 8 MB per XCD)
# and with the read loop's checker off (words folded, not verified): what the capacity argument of DESIGN 7.3 rests on, measured
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; out=gpurun_out/r06_j; mkdir -p $out
timeout 300 scripts/micro/l2_exchange.bin > $out/l2_exchange.txt 2>&1
for m in 6 7; do for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/l2x_${m}_$c; rm -rf $d
  (cd /tmp && TMPDIR=/tmp timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- "$R/scripts/micro/l2_exchange.bin" $m > /tmp/l2x.log 2>&1)
  python3 - $d $c $m >> $out/l2_exchange_pmc.txt <<'P'
import csv, glob, sys
tot = 0.0; n = 0
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == sys.argv[2] and "exch" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
print("mode %s %s: %.1f MB over %d launches (raw counter x 1024; FETCH_SIZE to be doubled per the guide)" % (sys.argv[3], sys.argv[2], tot * 1024 / 1e6, n))
P
done; grep "^mode" /tmp/l2x.log >> $out/l2_exchange_pmc.txt; done
cat $out/l2_exchange_pmc.txt; tail -12 $out/l2_exchange.txt
