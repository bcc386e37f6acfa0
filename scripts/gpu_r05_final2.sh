#!/bin/bash
# end of round 5, after the post-set experiments (sources changed in comments and variant-build guards only): the full GPU suite on the
# final build, the PMC traffic files re-measured so that they carry the final sources' hash, the driver's command, and the round's
# same-box A/B once more on this box (variants/r04eq = -DHS_DEC_STG_N=1 -DHS_BF_SWZ=0 -DHS_W256B_STG=0, run with HSIMAE_WGRAD_PLANAR=0)
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05_fin; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
for m in base large; do bash scripts/gpu_step_traffic.sh $m > /dev/null 2>&1; cp gpurun_out/step_traffic_$m.json $out/; cp gpurun_out/step_traffic_$m.json profiles/; done
bash scripts/gpu_step_traffic.sh huge > /dev/null 2>&1; cp gpurun_out/step_traffic_huge.json $out/step_traffic_huge_fp8.json; cp gpurun_out/step_traffic_huge.json profiles/step_traffic_huge_fp8.json
rm -rf gpurun_out/traffic_*
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_base.json 2> $out/bench_base.err; cut -c1-400 $out/bench_base.json
timeout 600 python bench.py --gpus 1 --force-ddp --steps 50 --warmup 10 --no-extras > $out/bench_base_ddp_path_1rank.json 2>/dev/null; cut -c1-200 $out/bench_base_ddp_path_1rank.json
b() { timeout 300 python bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
for rep in 1 2 3 4; do
  echo "round-5 build            $(b)" >> $out/ab.txt
  echo "round-4 equivalent build $(HSIMAE_WGRAD_PLANAR=0 HSIMAE_LIB=variants/r04eq/libhsimae_hip.so b)" >> $out/ab.txt
done
cat $out/ab.txt
