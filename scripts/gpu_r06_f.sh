#!/bin/bash
# round 6, call F: enc_mlp_bwd operand stores behind the du2 products (default) vs in front of them (r6f_early, same sources) vs the
# kernel before this round's changes to it (r6e_pf: row addresses as 64-bit pairs, biases from global) — Base and Large, kernel statistics
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; out=gpurun_out/r06_f; mkdir -p $out
for m in base large; do for rep in 1 2; do for L in hsimae_amd/libhsimae_hip.so variants/r6f_early/libhsimae_hip.so variants/r6e_pf/libhsimae_hip.so; do
  d=/tmp/ab_$RANDOM
  (cd /tmp && TMPDIR=/tmp HSIMAE_LIB="$R/$L" HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$R/bench.py" --model $m --steps 4 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
  echo "== $m $L" | tee -a $out/ab.txt
  python3 - "$d" <<'P' | tee -a $out/ab.txt
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:8]:
    if re.search("enc_mlp|wgrad|blk", r["Name"]):
        print("    %-48s %8.1f us x %s" % (re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:48], float(r["AverageNs"]) / 1e3, r["Calls"]))
P
  HSIMAE_LIB="$R/$L" timeout 300 python bench.py --model $m --steps 30 --warmup 8 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('    ms_per_step', d['ms_per_step'])" | tee -a $out/ab.txt
done; done; done
