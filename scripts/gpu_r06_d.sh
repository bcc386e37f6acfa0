#!/bin/bash
# round 6, call D: phase stamps of the decoder backward kernels (one / two query tiles in hand), the new parity tests once with their
# measured numbers printed (std = 0.08 at N = 4096, Huge fp8 at N = 1024 against the MX-operand oracle, schedule-record test), and the
# A/B old build (variants/r6_base) vs shipped default vs two query tiles
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; out=gpurun_out/r06_d; mkdir -p $out
for n in r6c_ph1 r6c_ph2; do PHASE_LIB=$R/variants/$n/libhsimae_hip.so timeout 300 python3 scripts/phase_timing.py 2>/dev/null | grep -A9 "^dec_bwd" > $out/phase_$n.txt; done
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sched.py -q -x -s -k "0.08 or huge_fp8_n1024 or encoder_only_forward" > $out/parity_new_tests.txt 2>&1; grep -E "^\[|passed|failed|Error|assert" $out/parity_new_tests.txt | tail -12
HSIMAE_LIB=$R/variants/r6c_nq2/libhsimae_hip.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x -k "fused_decoder or padded_key or tiny_model or odd_batches" > $out/parity_nq2.txt 2>&1; tail -1 $out/parity_nq2.txt
for rep in 1 2 3; do for L in variants/r6_base/libhsimae_hip.so hsimae_amd/libhsimae_hip.so variants/r6c_nq2/libhsimae_hip.so; do
  d=/tmp/ab_$RANDOM
  (cd /tmp && TMPDIR=/tmp HSIMAE_LIB="$R/$L" HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
  echo "== $L" | tee -a $out/ab.txt
  python3 - "$d" <<'P' | tee -a $out/ab.txt
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:10]:
    if re.search("dec_bwd", r["Name"]):
        print("    %-44s %8.1f us" % (re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:44], float(r["AverageNs"]) / 1e3))
P
  HSIMAE_LIB="$R/$L" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('    ms_per_step', d['ms_per_step'])" | tee -a $out/ab.txt
done; done
