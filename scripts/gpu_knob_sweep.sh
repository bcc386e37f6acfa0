cd "$GRAFT_REPO_ROOT"
run() { echo -n "$1: "; env $1 timeout 200 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
run A=1
run HSIMAE_WGRAD_WGS=384
run HSIMAE_WGRAD_WGS=640
run HSIMAE_WGRAD_WGS=768
run HSIMAE_WGRAD_WGS=1024
run HSIMAE_WGRAD_DS=4
run HSIMAE_BLK128_WGS=512
run HSIMAE_BLK128_SPW=1
run A=2
run HSIMAE_DEC_FWD_WGS=512
run HSIMAE_DEC_SPLIT=0
run HSIMAE_LNBWD_DMA=0
run A=3
