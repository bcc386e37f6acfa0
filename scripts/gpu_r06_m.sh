#!/bin/bash
# round 6, call M: enc_mlp_bwd<128,352> with x-hat / dY / 1/sigma kept in registers (no epilogue re-read: -113 MB per launch), two workgroups per CU
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; out=gpurun_out/r06_m; mkdir -p $out
HSIMAE_LIB=$R/variants/r6m_keep/libhsimae_hip.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x -k "encoder_mlp or tiny_model or odd_batches or c1_base48" > $out/parity.txt 2>&1; tail -1 $out/parity.txt
for rep in 1 2 3; do for L in hsimae_amd/libhsimae_hip.so variants/r6m_keep/libhsimae_hip.so; do
  d=/tmp/ab_$RANDOM
  (cd /tmp && TMPDIR=/tmp HSIMAE_LIB="$R/$L" HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
  echo "== $L" | tee -a $out/ab.txt
  python3 - "$d" <<'P' | tee -a $out/ab.txt
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:8]:
    if re.search("enc_mlp_bwd|wgrad_dma|blk128_bwd", r["Name"]):
        print("    %-52s %8.1f us x %s" % (re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:52], float(r["AverageNs"]) / 1e3, r["Calls"]))
P
  HSIMAE_LIB="$R/$L" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('    ms_per_step', d['ms_per_step'])" | tee -a $out/ab.txt
done; done
