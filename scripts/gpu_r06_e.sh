#!/bin/bash
# round 6, call E: the whole GPU suite on the library after the generation retirement (gen-1 attention, attn128_*, gemm_dma.hip gone),
# phase stamps at Large, A/B default vs MLP prefetch vs no packed fp32, default bench lines (base, large)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; out=gpurun_out/r06_e; mkdir -p $out
timeout 1500 python -m pytest tests -q -x -m gpu > $out/gpu_suite.txt 2>&1; tail -5 $out/gpu_suite.txt
PHASE_LIB=$R/variants/r6e_ph/libhsimae_hip.so MODEL=large timeout 300 python3 scripts/phase_timing.py > $out/phase_large.txt 2>/dev/null
PHASE_LIB=$R/variants/r6e_ph/libhsimae_hip.so timeout 300 python3 scripts/phase_timing.py > $out/phase_base.txt 2>/dev/null
for rep in 1 2; do for L in hsimae_amd/libhsimae_hip.so variants/r6e_pf/libhsimae_hip.so variants/r6e_nopk/libhsimae_hip.so; do
  d=/tmp/ab_$RANDOM
  (cd /tmp && TMPDIR=/tmp HSIMAE_LIB="$R/$L" HSIMAE_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline >/dev/null 2>&1)
  echo "== $L" | tee -a $out/ab.txt
  python3 - "$d" <<'P' | tee -a $out/ab.txt
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:12]:
    if not r["Name"].startswith(("void at::", "__amd")):
        print("    %-44s %8.1f us x %s" % (re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:44], float(r["AverageNs"]) / 1e3, r["Calls"]))
P
  HSIMAE_LIB="$R/$L" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('    ms_per_step', d['ms_per_step'])" | tee -a $out/ab.txt
done; done
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_base.json 2>$out/bench_base.err; tail -c 600 $out/bench_base.json
timeout 600 python bench.py --model large --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_large.json 2>/dev/null; python3 -c "import json; d=json.loads(open('$out/bench_large.json').read().strip().splitlines()[-1]); print('large ms', d['ms_per_step'])"
