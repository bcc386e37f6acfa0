#!/bin/bash
# round 6, call L (combinations, three repetitions): start stagger of the persistent decoder backward kernels re-swept on the round-6 kernels (HS_DEC_STG_ATTN 0 / 3 / 5 = default / 8,
# HS_DEC_STG_MLP 0 = default / 3 / 5 / 8; x 0.85 us between the two classes of workgroups)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; out=gpurun_out/r06_l; mkdir -p $out
for rep in 1 2 3; do for n in default r6l_a3m5 r6l_a4m6 r6l_a3m4; do
  L=variants/$n/libhsimae_hip.so; [ $n = default ] && L=hsimae_amd/libhsimae_hip.so
  d=/tmp/ab_$RANDOM
  (cd /tmp && TMPDIR=/tmp HSIMAE_LIB="$R/$L" HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
  echo "== $n" | tee -a $out/ab.txt
  python3 - "$d" <<'P' | tee -a $out/ab.txt
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:10]:
    if re.search("dec_bwd", r["Name"]):
        print("    %-48s %8.1f us x %s" % (re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:48], float(r["AverageNs"]) / 1e3, r["Calls"]))
P
done; done
for rep in 1 2 3; do for n in default r6l_a3m5 r6l_a4m6; do
  L=variants/$n/libhsimae_hip.so; [ $n = default ] && L=hsimae_amd/libhsimae_hip.so
  echo "== step $n" | tee -a $out/ab.txt
  HSIMAE_LIB="$R/$L" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('    ms_per_step', d['ms_per_step'])" | tee -a $out/ab.txt
done; done
