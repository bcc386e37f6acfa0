#!/usr/bin/env python3
"""HSIMAE pretraining fwd+bwd throughput on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one forward + backward of HSIMAE-Base over a per-GPU batch of 4096 synthetic 9x9x96 cubes already
resident in HBM (mask ratio 0.75), including the RCCL gradient all-reduce when N > 1.  Rank 0 prints ONE JSON
line; `value` is the whole-job patches/s.  `roofline` prices the kernel with the largest share of the step (HBM-bound)
against the HBM peak from live HIP-event timings (`roofline_wgrad`: the runner-up); `cpu_baseline` times the CPU oracle on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md chip table


def flops_per_sample(bands, D, depth, s_depth, Dd, dec_depth, lt, ll, hidden, dec_hidden):
    """Algorithmic fwd+bwd FLOPs per cube (SURVEY.md 8a closed form; recompute not credited)."""
    T = bands // 8
    TL, K = T * 9, lt * ll
    nfus = depth - s_depth if s_depth < 12 else 0
    PE = TL * 72 * D
    ENCl = K * (4 * D * D + 3 * D * hidden) * (2 * s_depth + nfus)
    ENCa = K * 2 * D * (s_depth * ll + s_depth * lt + nfus * K)
    DE = K * D * Dd
    DECl = TL * (4 * Dd * Dd + 3 * Dd * dec_hidden) * dec_depth
    DECa = TL * 2 * Dd * TL * dec_depth
    PRED = TL * Dd * 72
    fwd = 2 * (PE + ENCl + ENCa + DE + DECl + DECa + PRED)
    return 3 * fwd - 2 * PE


PEAK_HBM_GBS = 8000.0              # HBM3E spec, same guide
# HBM bytes per encoder-block wgrad launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE, guide's gfx950
# correction), see profiles/r01_pmc_wgrad.txt; None until measured
WGRAD_TRAFFIC_BYTES = 2 * 235753 * 1024 + 25368 * 1024   # 509 MB vs 460 MB algorithmic (profiles/r01_v15_pmc_hbm_traffic.txt)


# enc_mlp_bwd_kernel, same source: 2*116226 KB fetched + 370437 KB written = 617 MB vs 488 MB algorithmic (x1 / dY are
# re-read in the epilogue)
MLPBWD_TRAFFIC_BYTES = 2 * 116226 * 1024 + 370437 * 1024


def _timed_interleaved(launches, iters):
    """Average HIP-event duration of each launch in `launches`, issued round-robin back to back (A, B, A, B, ...) on one
    stream the way the backward of consecutive encoder blocks issues them: every launch finds L2 / Infinity Cache filled
    by the other kernel's ~0.5 GB, as inside a step, and there is no idle gap between launches (a replay of ONE kernel
    alone re-reads its own previous launch from cache and measured 15-20 % faster than the same kernel inside the step;
    single launches bracketed by an evicting fill measured 10-40 % slower because each event pair then includes the queue
    going idle)."""
    for fn in launches:
        for _ in range(2):
            fn()
    marks = []
    for _ in range(iters):
        row = [torch.cuda.Event(enable_timing=True) for _ in range(len(launches) + 1)]
        row[0].record()
        for j, fn in enumerate(launches):
            fn()
            row[j + 1].record()
        marks.append(row)
    torch.cuda.synchronize()
    return [sum(r[j].elapsed_time(r[j + 1]) for r in marks) / iters for j in range(len(launches))]


def dominant_kernel_roofline(model, N, K_tok, iters=20, return_launch=False, ms=None):
    """Live HIP-event timing of the kernel with the largest share of the step (profiles/r01_v11_kernel_stats_*:
    `enc_mlp_bwd_kernel`, 21 launches, 14-15 % of kernel time): the MLP-half backward of one ENCODER block at the
    workload's shape (M = N*K kept-token rows), through the C ABI on the current stream.  HBM-bound: algorithmic bytes
    per launch = every operand once = M * (x1 4d + dY 4d  read;  dx1 4d + u2 2d + dY_bf16 2d + dx1_bf16 2d + dh1|dh3
    2*2hp + g 2hp  written)."""
    from hsimae_amd import _lib, swiglu_hidden
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    d = model.dim
    h = swiglu_hidden(d, model.mlp_ratio)
    hp = (h + 31) // 32 * 32
    M = N * K_tok
    f32 = dict(dtype=torch.float32, device=dev)
    bf = dict(dtype=torch.bfloat16, device=dev)
    x1, dy = torch.randn(M, d, **f32), torch.randn(M, d, **f32) * 1e-3
    dx1 = torch.empty(M, d, **f32)
    u2, dyb, dx1b = (torch.empty(M, d, **bf) for _ in range(3))
    dh13, g = torch.empty(M, 2 * hp, **bf), torch.empty(M, hp, **bf)
    # packed weight images: any bf16 content of the right size is a valid image (timing only)
    w1, w3, w2T = (torch.randn(hp * d, **bf) * 0.05 for _ in range(3))
    w2, w13T = torch.randn(d * hp, **bf) * 0.05, torch.randn(d * 2 * hp, **bf) * 0.05
    n2w, n2b, b2 = torch.ones(d, **f32), torch.zeros(d, **f32), torch.zeros(d, **f32)
    b1, b3 = torch.zeros(hp, **f32), torch.zeros(hp, **f32)
    gw, gb = torch.zeros(d, **f32), torch.zeros(d, **f32)
    w = _lib.MlpWeights(n2w=n2w.data_ptr(), n2b=n2b.data_ptr(), w1b=b1.data_ptr(), w3b=b3.data_ptr(), w2b=b2.data_ptr(),
                        w1=w1.data_ptr(), w3=w3.data_ptr(), w2=w2.data_ptr(), w2T=w2T.data_ptr(), w13T=w13T.data_ptr(), hidden=h)
    s = torch.cuda.current_stream().cuda_stream

    def launch():
        _lib.check(lib.hsimae_enc_mlp_bwd(x1.data_ptr(), dy.data_ptr(), dx1.data_ptr(), u2.data_ptr(), dh13.data_ptr(),
                                          g.data_ptr(), dyb.data_ptr(), dx1b.data_ptr(), M, d, C.byref(w), gw.data_ptr(),
                                          gb.data_ptr(), None, None, s), "hsimae_enc_mlp_bwd")
    if return_launch:
        return launch
    if ms is None:
        ms = _timed_interleaved([launch], iters)[0]
    nbytes = float(M) * (4 * d + 4 * d + 4 * d + 2 * d + 2 * d + 2 * d + 2 * 2 * hp + 2 * hp)
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "enc_mlp_bwd_kernel<128,352> (encoder block: MLP-half backward, emits dx1 + the weight-gradient operands)",
            "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": MLPBWD_TRAFFIC_BYTES, "launch_ms": round(ms, 4),
            "bytes_per_launch": nbytes}


def wgrad_kernel_roofline(model, N, K_tok, iters=20, return_launch=False, ms=None):
    """Second-largest HBM-bound kernel (wgrad_dma_kernel, 23 launches, 12-13 % of kernel time): the batched
    weight-gradient launch of one ENCODER block (q, k, v, proj, w1, w3, w2) at the workload's shape, timed the same way.
    Algorithmic bytes per launch = every operand read once =
    M * (dqkv 3d*2 + u d*2 + dx1 d*2 + o d*2 + dh13 2hp*2 + u2 d*2 + dY d*2 + g hp*2)  (all operands bf16)."""
    from hsimae_amd import _lib, swiglu_hidden
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    d = model.dim
    h = swiglu_hidden(d, model.mlp_ratio)
    hp = (h + 31) // 32 * 32
    M = N * K_tok
    bf = dict(dtype=torch.bfloat16, device=dev)
    dqkv, u, o, u2 = (torch.randn(M, w, **bf) for w in (3 * d, d, d, d))
    dh13, g = torch.randn(M, 2 * hp, **bf), torch.randn(M, hp, **bf)
    G0, G1 = torch.randn(M, d, **bf), torch.randn(M, d, **bf)      # bf16 copies of dY / dx1 (emitted by enc_mlp_bwd)
    dW = [torch.zeros(n, k, device=dev) for n, k in ((d, d),) * 4 + ((h, d),) * 2 + ((d, h),)]
    db = [torch.zeros(w.shape[0], device=dev) for w in dW]
    wp = _lib.WgradParams()
    spec = [(dqkv.data_ptr(), 0, 3 * d, u, d, d, d), (dqkv.data_ptr() + 2 * d, 0, 3 * d, u, d, d, d),
            (dqkv.data_ptr() + 4 * d, 0, 3 * d, u, d, d, d), (G1.data_ptr(), 0, d, o, d, d, d),
            (dh13.data_ptr(), 0, 2 * hp, u2, d, h, d), (dh13.data_ptr() + 2 * hp, 0, 2 * hp, u2, d, h, d),
            (G0.data_ptr(), 0, d, g, hp, d, h)]
    tiles = 0
    for i, (dO, f32, ldo, A, lda, n, k) in enumerate(spec):
        wp.t[i] = _lib.WgradTask(dO=dO, dO_f32=f32, ldo=ldo, A=A.data_ptr(), lda=lda, N=n, K=k, dW=dW[i].data_ptr(),
                                 ldw=k, db=db[i].data_ptr())
        tiles += ((n + 127) // 128) * ((k + 127) // 128)
    wp.ntasks, wp.M, wp.msplit = len(spec), M, lib.hsimae_wgrad_msplit(tiles, M)    # what hsimae_backward launches
    s = torch.cuda.current_stream().cuda_stream
    def launch():
        _lib.check(lib.hsimae_wgrad(C.byref(wp), s))
    if return_launch:
        launch.keep = (wp, dqkv, u, o, u2, dh13, g, G0, G1, dW, db)
        return launch
    if ms is None:
        ms = _timed_interleaved([launch], iters)[0]
    nbytes = float(M) * (3 * d * 2 + d * 2 + d * 2 + d * 2 + 2 * hp * 2 + d * 2 + d * 2 + hp * 2)
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "wgrad_dma_kernel (encoder block: dW/db of q,k,v,proj,w1,w3,w2 in one launch)",
            "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": WGRAD_TRAFFIC_BYTES, "launch_ms": round(ms, 4),
            "bytes_per_launch": nbytes}


def optimizer_step_ms(model, iters=10):
    """Not part of the metric (fwd+bwd only): the reference loop's optimizer.step() (Model_Pretraining.py:102) as
    stock torch AdamW over 535 tensors vs the one-launch FusedAdamW + packed-weight refresh."""
    from hsimae_amd import FusedAdamW
    nd = ["bias", "norm"]
    groups = [{"params": [p for n, p in model.named_parameters() if not any(k in n for k in nd)], "weight_decay": 5e-2},
              {"params": [p for n, p in model.named_parameters() if any(k in n for k in nd)], "weight_decay": 0.0}]
    ref = torch.optim.AdamW(groups, lr=1e-9, weight_decay=5e-2, betas=(0.9, 0.95))
    fused = FusedAdamW(model, lr=1e-9, weight_decay=5e-2, betas=(0.9, 0.95))
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, opt in (("torch_adamw", ref), ("fused_adamw_plus_repack", fused)):
        for _ in range(2):
            opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            opt.step()
            if opt is fused:
                model._ensure_packed(stream)
        torch.cuda.synchronize()
        res[name] = round((time.perf_counter() - t0) / iters * 1e3, 3)
    return res


def input_pipeline_ms(bands, N, iters=10):
    """Not part of the metric (inputs are resident when timing starts): SURVEY 8f row N2, one batch of N cubes
    assembled on the device from HBM-resident synthetic scenes (hsimae_cube_gather through hsimae_amd.data).
    HBM-bound byte kernel: algorithmic bytes = read + write 324*bands B per cube."""
    import numpy as np
    from hsimae_amd.data import HSIdataset4PT
    rng = np.random.default_rng(0)
    scenes = [rng.random((145, 145, bands), dtype=np.float32) for _ in range(4)]         # Indian-Pines-sized scenes
    cut = np.array([(0, h, w, s, 1, 0) for s in range(4) for h in range(0, 136, 3) for w in range(0, 136, 3)], dtype=np.int16)
    ds = HSIdataset4PT([scenes, cut], train=True)
    idx = rng.integers(0, len(cut), N).tolist()
    flips = rng.integers(0, 4, N).astype(np.uint8)
    for _ in range(2):
        ds.gather(idx, flips)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ds.gather(idx, flips)
    e1.record()
    torch.cuda.synchronize()
    dev_ms = e0.elapsed_time(e1) / iters               # includes the index / flip uploads of each call
    t0 = time.perf_counter()
    for _ in range(iters):
        ds.batch(idx)                                   # + 2 python random() draws per cube, as the reference consumes them
    torch.cuda.synchronize()
    call_ms = (time.perf_counter() - t0) / iters * 1e3
    nbytes = 2.0 * N * 81 * bands * 4
    return {"device_ms_per_batch": round(dev_ms, 4), "host_call_ms_per_batch": round(call_ms, 3),
            "cubes_per_s": round(N / (call_ms * 1e-3), 0), "algorithmic_GBps": round(nbytes / (dev_ms * 1e-3) / 1e9, 1)}


def cpu_baseline(bands, n_sample=64, steps=4):
    """The CPU oracle (a port of the reference's algorithm, validated against it) on the host cores."""
    from oracle import hsimae_oracle as O
    # torch's CPU ops on these small shapes stop scaling (and then collapse) past a few dozen threads:
    # use at most 32 of the host's cores and report that number.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    cfg = O.OracleConfig(bands=bands)
    state = O.init_state(cfg, seed=0)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(n_sample, 1, bands, 9, 9, generator=g)
    n1, n2 = torch.rand(n_sample, cfg.T, generator=g), torch.rand(n_sample, 9, generator=g)
    t0 = time.perf_counter()
    O.forward_backward(state, cfg, x, n1.numpy(), n2.numpy(), 3, 9)            # warm-up (also calibrates the sample)
    warm = time.perf_counter() - t0
    steps = max(1, min(steps, int(12.0 / max(warm, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        O.forward_backward(state, cfg, x, n1.numpy(), n2.numpy(), 3, 9)
    dt = time.perf_counter() - t0
    return {"value": round(n_sample * steps / dt, 2), "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": f"oracle fwd+bwd fp32, HSIMAE-Base 9x9x{bands}, batch {n_sample}, {steps} steps after 1 warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch (weak scaling)")
    ap.add_argument("--model", default="base", choices=["base", "large"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-ddp", action="store_true", help="run the RCCL gradient reducer even with one rank (test)")
    args = ap.parse_args()

    import torch.distributed as dist
    from hsimae_amd import HSIMAE, swiglu_hidden

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_ddp = world > 1 or args.force_ddp
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    bands, D = 96, (128 if args.model == "base" else 256)
    torch.manual_seed(0)
    model = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=bands, b_patch_size=8, embed_dim=D, depth=12,
                   num_heads=D // 16, s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8,
                   norm_pix_loss=True, trunc_init=True).to(dev)
    if use_ddp:
        model.enable_data_parallel()
    random.seed(0)                                    # same (len_t, len_l) sequence on every rank
    torch.manual_seed(1234 + rank)
    N = args.batch
    imgs = torch.rand(N, 1, bands, 9, 9, device=dev)  # synthetic cubes, resident in HBM

    def step():
        model.zero_grad(set_to_none=True)
        loss, _, _ = model(imgs, mask_ratio=0.75)
        loss.backward()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    last_loss = float(loss.item())

    if rank == 0:
        h, hd = swiglu_hidden(D, 4.0), swiglu_hidden(64, 4.0)
        fl = flops_per_sample(bands, D, 12, 9, 64, 8, 3, 9, h, hd)
        value = world * N * args.steps / dt
        step_tflops = value * fl / 1e12 / world
        out = {
            "metric": "HSI patches/sec (9x9x96, mask 75%) pretrain fwd+bwd", "value": round(value, 1),
            "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"HSIMAE-{args.model.capitalize()} pretrain fwd+bwd, 9x9x96 cubes, per-GPU batch {N}, "
                                   f"mask 0.75, bf16 MFMA operands / fp32 accumulate+residual",
                       "per_gpu_batch": N, "global_batch": N * world, "parallelism": f"dp{world}"},
            "per_gpu": round(value / world, 1), "loss": round(last_loss, 6),
            "step_algorithmic_tflops_per_gpu": round(step_tflops, 2),
            "step_frac_of_bf16_peak": round(step_tflops / PEAK_BF16_TFLOPS, 4),
            "gflop_per_patch": round(fl / 1e9, 4),
        }
        if D == 128:
            la = dominant_kernel_roofline(model, N, 27, return_launch=True)
            lb = wgrad_kernel_roofline(model, N, 27, return_launch=True)
            ms_a, ms_b = _timed_interleaved([la, lb], 20)
            out["roofline"] = dominant_kernel_roofline(model, N, 27, ms=ms_a)
            out["roofline_wgrad"] = wgrad_kernel_roofline(model, N, 27, ms=ms_b)
        else:                                   # wider encoders run layer-at-a-time: the weight-gradient launch leads there
            out["roofline"] = wgrad_kernel_roofline(model, N, 27)
            out["roofline"]["traffic"] = None   # PMC traffic was collected at D = 128 only
        out["optimizer_step_ms"] = optimizer_step_ms(model)
        out["input_pipeline"] = input_pipeline_ms(bands, N)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(bands)
        print(json.dumps(out), flush=True)
    if use_ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
