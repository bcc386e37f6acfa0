#!/bin/bash
# round 6, call C: the decoder backward kernels without spills / vmcnt(0) drains (uniform base + lane byte offset, geometry re-derived
# per phase, bounds-checked buffer accesses): parity of each variant, then same-box A/B against the build before the change (r6_base)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; out=gpurun_out/r06_c; mkdir -p $out
for n in r6b_new r6b_nq2pf; do
HSIMAE_LIB=$R/variants/$n/libhsimae_hip.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x -k "fused_decoder or padded_key or tiny_model or odd_batches" > $out/parity_$n.txt 2>&1; tail -2 $out/parity_$n.txt
done
libs="r6_base r6b_new r6b_nq2 r6b_pf r6b_nq2pf r6b_noregeo r6b_nq2pfprio"
for rep in 1 2; do for n in $libs; do
  L=variants/$n/libhsimae_hip.so; d=/tmp/ab_$RANDOM
  (cd /tmp && TMPDIR=/tmp HSIMAE_LIB="$R/$L" HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
  echo "== $n" | tee -a $out/ab.txt
  python3 - "$d" <<'P' | tee -a $out/ab.txt
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:10]:
    if re.search("dec_bwd", r["Name"]):
        print("    %-44s %8.1f us" % (re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:44], float(r["AverageNs"]) / 1e3))
P
  HSIMAE_LIB="$R/$L" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('    ms_per_step', d['ms_per_step'])" | tee -a $out/ab.txt
done; done
