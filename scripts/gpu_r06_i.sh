#!/bin/bash
# round 6, call I: dec_bwd_mlp with the next sample's rows requested at the start of the last chunk's weight-gradient phase (HS_DEC_MLP_PREFETCH=2)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; out=gpurun_out/r06_i; mkdir -p $out
HSIMAE_LIB=$R/variants/r6i_pf2/libhsimae_hip.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x -k "fused_decoder or padded_key or tiny_model or odd_batches" > $out/parity.txt 2>&1; tail -1 $out/parity.txt
for rep in 1 2 3; do for L in hsimae_amd/libhsimae_hip.so variants/r6i_pf2/libhsimae_hip.so; do
  d=/tmp/ab_$RANDOM
  (cd /tmp && TMPDIR=/tmp HSIMAE_LIB="$R/$L" HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
  echo "== $L" | tee -a $out/ab.txt
  python3 - "$d" <<'P' | tee -a $out/ab.txt
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:10]:
    if re.search("dec_bwd", r["Name"]):
        print("    %-48s %8.1f us x %s" % (re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:48], float(r["AverageNs"]) / 1e3, r["Calls"]))
P
  HSIMAE_LIB="$R/$L" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('    ms_per_step', d['ms_per_step'])" | tee -a $out/ab.txt
done; done
