#!/bin/bash
# round 6, call A: (1) the L2-exchange gate (scripts/micro/l2_exchange.hip) with its FETCH_SIZE / WRITE_SIZE passes,
# (2) parity of the two-query-tile attention core, (3) same-box A/B of the de-phasing variants (variants/r6_*, built by
# scripts/build_variant.py): kernel statistics single-stream + two-stream step time, two alternating repetitions
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; out=gpurun_out/r06_a; mkdir -p $out
timeout 120 scripts/micro/l2_exchange.bin > $out/l2_exchange.txt 2>&1
for m in 1 2 3 4 5; do for c in FETCH_SIZE WRITE_SIZE; do
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
HSIMAE_LIB=$R/variants/r6_nq2/libhsimae_hip.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x -k "fused_decoder or padded_key or tiny_model" > $out/parity_nq2.txt 2>&1; tail -3 $out/parity_nq2.txt
libs="r6_base r6_prio1 r6_prio2 r6_nq2 r6_nq2prio r6_stg4 r6_blkprio"
for rep in 1 2; do for n in $libs; do
  L=variants/$n/libhsimae_hip.so; d=/tmp/ab_$RANDOM
  (cd /tmp && TMPDIR=/tmp HSIMAE_LIB="$R/$L" HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
  echo "== $n" | tee -a $out/ab.txt
  python3 - "$d" <<'P' | tee -a $out/ab.txt
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:10]:
    if re.search("dec_bwd|blk128|dec_attn", r["Name"]):
        print("    %-44s %8.1f us" % (re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:44], float(r["AverageNs"]) / 1e3))
P
  HSIMAE_LIB="$R/$L" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('    ms_per_step', d['ms_per_step'])" | tee -a $out/ab.txt
done; done
