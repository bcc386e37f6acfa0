#!/usr/bin/env python3
"""HSIMAE pretraining fwd+bwd throughput on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N --steps K --warmup W] [--model base|large|huge] [--precision bf16|fp8]

`--gpus N` with N > 1 starts its own N ranks (`python -m torch.distributed.run --nproc-per-node N ... bench.py ...`) from a
parent process that never touches the GPU; under an external launcher (RANK / WORLD_SIZE in the environment) the
process is one of those ranks.  A step = one forward + backward of HSIMAE over a per-GPU batch of synthetic cubes already
resident in HBM (mask ratio 0.75), including the RCCL gradient all-reduce when N > 1.  Rank 0 prints ONE JSON line;
`value` is the whole-job patches/s over the wall-clock bracket (barrier + synchronize on both sides, MAX over ranks).

  roofline             SURVEY 8(d): the path is MFMA-bound; achieved = algorithmic fwd+bwd TFLOP/s per GPU of the step
                       (no recompute credit) against the dense bf16 (fp8: MX-scaled fp8) MFMA peak; `traffic` = HBM bytes
                       per step from the rocprofv3 PMC passes recorded in profiles/ (2*FETCH_SIZE + WRITE_SIZE).
  encoder_mfma_frac    encoder fwd+bwd alone (hsimae_encode + hsimae_encode_backward), HIP events, same peak.
  roofline_kernel      the kernel with the largest share of the step, replayed through the C ABI with HIP events:
                       its algorithmic FLOPs against the MFMA peak, and (roofline_kernel_hbm) its bytes against HBM.
  roofline_decoder     the decoder group (hsimae_decode / hsimae_decode_backward) with HIP events; roofline_decoder_block = ONE fused
                       decoder Block replayed through hsimae_dec_block_fwd / _bwd (forward pair; two backward kernels + reduce).
  step_ms              HIP-event duration of every timed step: median / p10 / p90, and per (len_t, len_l) grid.
  cpu_baseline         the CPU oracle on a bounded sample of the same workload (C1 and C2 shapes), on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import random
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_FP8_TFLOPS = 5000.0           # dense MX-scaled fp8 MFMA, same table
PEAK_HBM_GBS = 8000.0              # HBM3E spec, same guide

MODELS = {   # name: (bands, embed_dim, heads, default per-GPU batch)   -- depths 12 / 9, decoder [8, 64] (Model_Pretraining.py:130-131)
    "base": (96, 128, 8, 4096),
    "large": (96, 256, 16, 4096),
    "huge": (192, 512, 32, 1024),      # "Huge" is this repo's definition: embed_dim 512, 32 heads
}


def flops_per_sample(bands, D, depth, s_depth, Dd, dec_depth, lt, ll, hidden, dec_hidden, parts=False):
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
    total = 3 * fwd - 2 * PE
    if parts:
        enc = 3 * 2 * (ENCl + ENCa) + 2 * 2 * PE          # encoder stacks + patch embedding (no input gradient)
        return total, enc
    return total


def profile_traffic(model_name):
    """HBM bytes per step from the PMC summary committed under profiles/ (None when no pass was recorded for this model)."""
    path = os.path.join(ROOT, "profiles", f"step_traffic_{model_name}.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


# --------------------------------------------------------------------------- multi-rank launch
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    """GPUs this process would see, WITHOUT loading the HIP runtime: KFD topology nodes that have SIMDs, cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  None when the topology cannot be read (the ranks then find out)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def launch_ranks(args, argv):
    """Parent of a `--gpus N` run: start N ranks with torch.distributed.run and relay their output.  This process never
    touches the HIP runtime (GPUs are counted from the KFD topology in sysfs) and never re-execs."""
    if not args.dry_run:
        have = visible_gpus()
        if have is not None and have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


# --------------------------------------------------------------------------- live kernel replays (HIP events)
def _timed_interleaved(launches, iters):
    """Average HIP-event duration of each launch in `launches`, issued round-robin back to back (A, B, A, B, ...) on one
    stream the way the backward of consecutive encoder blocks issues them: every launch finds L2 / Infinity Cache filled
    by the other kernel's ~0.5 GB, as inside a step, and there is no idle gap between launches."""
    import torch
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


def mlp_bwd_launch(model, M):
    """One launch of the encoder block's MLP-half backward (`hsimae_enc_mlp_bwd`) at the workload's row count."""
    import torch
    from hsimae_amd import _lib, swiglu_hidden
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    d = model.dim
    h = swiglu_hidden(d, model.mlp_ratio)
    hp = (h + 31) // 32 * 32
    f32 = dict(dtype=torch.float32, device=dev)
    bf = dict(dtype=torch.bfloat16, device=dev)
    x1, dy = torch.randn(M, d, **f32), torch.randn(M, d, **f32) * 1e-3
    dx1 = torch.empty(M, d, **f32)
    u2, dyb, dx1b = (torch.empty(M, d, **bf) for _ in range(3))
    hp64 = (hp + 63) // 64 * 64                              # the schedule's operand layout: 64-column planes of M + 48 rows when M % 32 == 0
    planar = M + 48 if M % 32 == 0 else 0
    dh13, g = torch.empty(M + 48, 2 * hp64, **bf), torch.empty(M + 48, hp64, **bf)
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
                                          gb.data_ptr(), None, None, planar, s), "hsimae_enc_mlp_bwd")
    launch.keep = (x1, dy, dx1, u2, dyb, dx1b, dh13, g, w1, w3, w2T, w2, w13T, n2w, n2b, b2, b1, b3, gw, gb, w)
    launch.flops = float(M) * 3 * 2 * d * h                 # dg = dY W2, du2 = dh1 W1 + dh3 W3 (the h1 / h3 recompute is not credited)
    launch.design_bytes = float(M) * (4 * d + 4 * d + 4 * d + 2 * d + 2 * d + 2 * d + 2 * 2 * hp + 2 * hp)
    launch.compulsory_bytes = float(M) * (4 * d + 4 * d + 4 * d)     # x1, dY in; dx1 out
    launch.name = f"enc_mlp_bwd_kernel<{d},{hp}> (encoder block: MLP-half backward; emits dx1 + the weight-gradient operands)"
    return launch


def wgrad_launch(model, M):
    """One launch of the encoder block's batched weight-gradient kernel (q, k, v, proj, w1, w3, w2) at the workload's rows."""
    import torch
    from hsimae_amd import _lib, swiglu_hidden
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    d = model.dim
    h = swiglu_hidden(d, model.mlp_ratio)
    hp = (h + 31) // 32 * 32
    bf = dict(dtype=torch.bfloat16, device=dev)
    dqkv, u, o, u2 = (torch.randn(M, w, **bf) for w in (3 * d, d, d, d))
    # the schedule's operand layout: dh1 | dh3 and g as 64-column planes of M + 48 rows when M % 32 == 0, row-major otherwise
    hp64, R = (hp + 63) // 64 * 64, (M + 48 if M % 32 == 0 else 0)
    dh13 = torch.randn((R or M) * 2 * (hp64 if R else hp), **bf)
    g = torch.randn((R or M) * (hp64 if R else hp), **bf)
    G0, G1 = torch.randn(M, d, **bf), torch.randn(M, d, **bf)      # bf16 copies of dY / dx1 (emitted by enc_mlp_bwd)
    dW = [torch.zeros(n, k, device=dev) for n, k in ((d, d),) * 4 + ((h, d),) * 2 + ((d, h),)]
    db = [torch.zeros(w.shape[0], device=dev) for w in dW]
    wp = _lib.WgradParams()
    spec = [(dqkv.data_ptr(), 0, 3 * d, u, d, d, d), (dqkv.data_ptr() + 2 * d, 0, 3 * d, u, d, d, d),
            (dqkv.data_ptr() + 4 * d, 0, 3 * d, u, d, d, d), (G1.data_ptr(), 0, d, o, d, d, d),
            (dh13.data_ptr(), 0, hp64 if R else 2 * hp, u2, d, h, d),
            (dh13.data_ptr() + (2 * hp64 * R if R else 2 * hp), 0, hp64 if R else 2 * hp, u2, d, h, d),
            (G0.data_ptr(), 0, d, g, hp64 if R else hp, d, h)]
    tiles = 0
    for i, (dO, f32, ldo, A, lda, n, k) in enumerate(spec):
        wp.t[i] = _lib.WgradTask(dO=dO, dO_f32=f32, ldo=ldo, A=A.data_ptr(), lda=lda, N=n, K=k, dW=dW[i].data_ptr(),
                                 ldw=k, db=db[i].data_ptr(), dO_plane_rows=R if i in (4, 5) else 0, A_plane_rows=R if i == 6 else 0)
        tiles += ((n + 127) // 128) * ((k + 127) // 128)
    wp.ntasks, wp.M, wp.msplit = len(spec), M, lib.hsimae_wgrad_msplit(tiles, M)    # what hsimae_backward launches
    s = torch.cuda.current_stream().cuda_stream

    def launch():
        _lib.check(lib.hsimae_wgrad(C.byref(wp), s))
    launch.keep = (wp, dqkv, u, o, u2, dh13, g, G0, G1, dW, db)
    launch.flops = float(M) * 2 * (4 * d * d + 3 * d * h)
    launch.design_bytes = float(M) * (3 * d * 2 + d * 2 + d * 2 + d * 2 + 2 * hp * 2 + d * 2 + d * 2 + hp * 2)
    launch.compulsory_bytes = launch.design_bytes            # every operand read once; the operands themselves are a design choice
    launch.name = "wgrad_dma_kernel (encoder block: dW/db of q,k,v,proj,w1,w3,w2 in one launch)"
    return launch


def decoder_block_replay(model, N, Ts, peak_tflops, iters=10):
    """One fused decoder Block through the C ABI (`hsimae_dec_block_fwd` / `_bwd`: the attention-half + MLP-half forward pair, the
    two persistent backward kernels + the slab reduce) on synthetic rows, HIP events, algorithmic FLOPs.  None where the fused
    decoder does not cover the sequence length (216 tokens)."""
    import torch
    from hsimae_amd import _lib, swiglu_hidden
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    d, h = 64, swiglu_hidden(64, model.mlp_ratio)
    hp = (h + 31) // 32 * 32
    M = N * Ts
    f32 = dict(dtype=torch.float32, device=dev)
    bf = dict(dtype=torch.bfloat16, device=dev)
    x, dy = torch.randn(M, d, **f32), torch.randn(M, d, **f32) * 1e-3
    x1, x2, dx1, dx = (torch.empty(M, d, **f32) for _ in range(4))
    o, lse = torch.empty(M, d, **bf), torch.empty(M, 8, **f32)
    vec = {k: torch.zeros(n, **f32) for k, n in (("n1b", d), ("bqkv", 3 * d), ("pb", d), ("n2b", d), ("w1b", hp), ("w3b", hp), ("w2b", d))}
    vec.update(n1w=torch.ones(d, **f32), n2w=torch.ones(d, **f32))
    img = {k: torch.randn(n, **bf) * 0.05 for k, n in (("qkv", 3 * d * d), ("p", d * d), ("w1", hp * d), ("w3", hp * d), ("w2", d * hp), ("w2T", hp * d))}
    mas = {k: torch.randn(r, d, **f32) * 0.05 for k, r in (("qf", d), ("kf", d), ("vf", d), ("pf", d), ("w1f", h), ("w3f", h))}
    W = _lib.DecBlockWeights(hidden=h, **{k: v.data_ptr() for k, v in {**vec, **img, **mas}.items()})
    shapes = dict(n1w=(d,), n1b=(d,), qw=(d, d), qb=(d,), kw=(d, d), kb=(d,), vw=(d, d), vb=(d,), pw=(d, d), pb=(d,), n2w=(d,), n2b=(d,),
                  w1w=(h, d), w1b=(h,), w2w=(d, h), w2b=(d,), w3w=(h, d), w3b=(h,))
    G_ = {k: torch.zeros(s, **f32) for k, s in shapes.items()}
    Gs = _lib.DecBlockGrads(**{k: v.data_ptr() for k, v in G_.items()})
    slab = torch.empty(_lib.load().hsimae_dec_block_slab_floats(), **f32)
    s = torch.cuda.current_stream().cuda_stream

    def fwd():
        return lib.hsimae_dec_block_fwd(C.byref(W), x.data_ptr(), x1.data_ptr(), x2.data_ptr(), o.data_ptr(), lse.data_ptr(), N, Ts, 1, s)

    def bwd():
        return lib.hsimae_dec_block_bwd(C.byref(W), C.byref(Gs), x.data_ptr(), x1.data_ptr(), dy.data_ptr(), dx1.data_ptr(), dx.data_ptr(),
                                        o.data_ptr(), lse.data_ptr(), N, Ts, slab.data_ptr(), s)
    if fwd() != 0:
        return None
    _lib.check(bwd(), "hsimae_dec_block_bwd")
    tf, tb = _timed_interleaved([fwd, bwd], iters)
    fl_f = float(N) * (Ts * 2.0 * (4 * d * d + 3 * d * h) + 2 * 2.0 * Ts * Ts * d)      # linears + QK^T and PV
    keep = (x, dy, x1, x2, dx1, dx, o, lse, vec, img, mas, G_, slab, W, Gs)
    del keep
    return {"bound": "mfma", "what": "one fused decoder Block replayed through hsimae_dec_block_fwd / _bwd (forward: attention-half + "
            "MLP-half kernels; backward: dec_bwd_mlp + dec_bwd_attn + dec_dw_reduce), HIP events, algorithmic FLOPs (backward = 2 x forward)",
            "fwd_us": round(tf * 1e3, 1), "bwd_us": round(tb * 1e3, 1), "peak": peak_tflops, "unit": "TFLOP/s",
            "fwd_achieved": round(fl_f / (tf * 1e-3) / 1e12, 1), "bwd_achieved": round(2 * fl_f / (tb * 1e-3) / 1e12, 1),
            "frac": round(3 * fl_f / ((tf + tb) * 1e-3) / 1e12 / peak_tflops, 4), "flops_fwd": fl_f,
            "compulsory_bytes": {"fwd": float(M) * (4 * 4 * d + 2 * 2 * d + 2 * 32), "bwd": float(M) * (6 * 4 * d + 2 * d + 32)}}


def _build_info():
    from hsimae_amd import _lib
    from hsimae_amd.build import kernel_source_hash
    b = _lib.build_info()
    return {"variant": b["variant"], "variant_bits": b["variant_bits"], "default_flags": b["default_flags"], "flags_hash": b["flags_hash"],
            "kernel_source_hash": b["kernel_source_hash"], "matches_sources": b["kernel_source_hash"] == kernel_source_hash(),
            "lib": os.path.relpath(b["path"], os.path.dirname(os.path.abspath(__file__)))}


class _Watchdog:
    """Rank 0's guarantee of ONE JSON line: `finish()` prints the complete line; if it has not been called `seconds` after the timed
    region ended, a timer thread prints what the line holds by then (the contract fields are all there) and ends the process."""

    def __init__(self, seconds, line):
        import threading
        self.line, self.lock, self.done = line, threading.Lock(), False
        self.timer = None
        if seconds and seconds > 0:
            self.timer = threading.Timer(seconds, self._expired, args=(seconds,))
            self.timer.daemon = True
            self.timer.start()

    def _emit(self):
        C.CDLL(None).fflush(None)          # RCCL's banner sits in the C stdio buffer: flush it so the JSON is the LAST line
        print(json.dumps(self.line), flush=True)

    def _expired(self, seconds):
        with self.lock:
            if self.done:
                return
            self.done = True
            # the main thread may still be filling the dict (a slow leg, not a hung one): snapshot it, and fall back to the contract
            # fields if even that keeps failing — a line must come out
            snap = None
            for _ in range(5):
                try:
                    snap = dict(self.line)
                    snap["watchdog"] = (f"the legs after the timed region did not finish within {seconds:g} s: line printed without them "
                                        f"(have: {sorted(k for k in snap if k.startswith(('comm', 'dp_', 'roofline_', 'cpu_')))})")
                    text = json.dumps(snap)
                    break
                except RuntimeError:
                    snap = None
            if snap is None:
                keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                        "dtype", "data")
                text = json.dumps({**{k: self.line.get(k) for k in keys}, "watchdog": f"legs after the timed region hung ({seconds:g} s); partial line"})
            C.CDLL(None).fflush(None)
            print(text, flush=True)
        # a process that has used the GPU and gives up exits NON-ZERO (the line above is incomplete, other ranks may still be
        # blocked in a collective); never re-exec
        os._exit(3)

    def finish(self):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.timer is not None:
                self.timer.cancel()
            self._emit()


def kernel_rooflines(model, M, peak_tflops, iters=20):
    """The two kernels with the largest share of the step, replayed interleaved; MFMA and HBM pricing of each."""
    from hsimae_amd import _lib
    lib = _lib.load()
    launches = [wgrad_launch(model, M)]
    try:
        la = mlp_bwd_launch(model, M)
        la()                                    # HSIMAE_EUNSUPPORTED for widths without the fused MLP-half kernels
        launches.insert(0, la)
    except RuntimeError:
        pass
    ms = _timed_interleaved(launches, iters)
    nblk = 2 * model.s_depth + (max(0, model.depth - model.s_depth) if model.s_depth < 12 else 0)
    per_step = {id(la): (nblk if la is not launches[-1] else nblk + 2) for la in launches}   # the batched launch also serves embed / head
    order = sorted(zip(launches, ms), key=lambda lt: -lt[1] * per_step[id(lt[0])])           # largest share of the step first
    out = []
    for la, t in order:
        tf = la.flops / (t * 1e-3) / 1e12
        gb = la.design_bytes / (t * 1e-3) / 1e9
        out.append(({"bound": "mfma", "kernel": la.name, "achieved": round(tf, 1), "peak": peak_tflops, "unit": "TFLOP/s",
                     "frac": round(tf / peak_tflops, 4), "flops_per_launch": la.flops, "launch_ms": round(t, 4),
                     "launches_per_step": per_step[id(la)]},
                    {"bound": "hbm", "kernel": la.name, "achieved": round(gb, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(gb / PEAK_HBM_GBS, 4), "bytes_per_launch": la.design_bytes,
                     "compulsory_bytes_per_launch": la.compulsory_bytes, "launch_ms": round(t, 4)}))
    return out


def encoder_only_ms(model, imgs, iters=10):
    """HIP-event time of the encoder alone: hsimae_encode (patch embedding, masking, the three stacks, `norm`) +
    hsimae_encode_backward from d(latent), per batch."""
    import torch
    from hsimae_amd import _lib
    lib = _lib.load()
    cfg = model._config()
    dev = imgs.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    nocb = _lib.BUCKET_CB(0)
    times = []
    dlat = None
    for i in range(iters + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            _, _, _, st = model._run_forward(imgs, 0.75, None, None, want_latent=True, encoder_only=True)
        if dlat is None or dlat.shape != st["latent"].shape:
            dlat = torch.randn_like(st["latent"]) * 1e-3
        model._flat_scratch.zero_()
        _lib.check(lib.hsimae_encode_backward(C.byref(cfg), C.byref(st["io"]), dlat.data_ptr(),
                                              model._flat_scratch.data_ptr(), nocb, None, stream), "hsimae_encode_backward")
        e1.record()
        times.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in times[2:])
    return ts[len(ts) // 2]


def decoder_only_ms(model, imgs, iters=10):
    """HIP-event times of the decoder alone, forward and backward separately: hsimae_decode (decoder_embed, assembly, the
    decoder blocks, decoder_norm + decoder_pred) and hsimae_decode_backward from d(pred), per batch."""
    import torch
    with torch.no_grad():
        latent, mask, ids_restore, _ = model.forward_encoder(imgs, 0.75)
    fwd, bwd = [], []
    dpred = None
    for i in range(iters + 2):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        pred, st = model._run_decode(latent, ids_restore)
        e1.record()
        if dpred is None:
            dpred = torch.randn_like(pred) * 1e-3
        model._decode_backward(st, dpred)
        e2.record()
        fwd.append((e0, e1)); bwd.append((e1, e2))
    torch.cuda.synchronize()
    med = lambda ev: sorted(a.elapsed_time(b) for a, b in ev[2:])[len(ev[2:]) // 2]
    model.zero_grad(set_to_none=True)
    return med(fwd), med(bwd)


def optimizer_step_ms(model, imgs, iters=10):
    """Not part of the metric (fwd+bwd only): the reference loop's optimizer.step() (Model_Pretraining.py:102) as
    stock torch AdamW over 535 tensors vs the one-launch FusedAdamW + packed-weight refresh.  Each leg
    starts from a zero_grad + forward + backward, so every trainable parameter has a gradient (round 3 timed both after
    zero_grad(set_to_none=True): torch skipped all 535 tensors and the fused optimizer took its missing-gradient path); the
    `iters` steps are then issued back to back — the gradients stay attached — between two synchronizes."""
    import torch
    from hsimae_amd import FusedAdamW
    nd = ["bias", "norm"]
    groups = [{"params": [p for n, p in model.named_parameters() if not any(k in n for k in nd)], "weight_decay": 5e-2},
              {"params": [p for n, p in model.named_parameters() if any(k in n for k in nd)], "weight_decay": 0.0}]
    ref = torch.optim.AdamW(groups, lr=1e-9, weight_decay=5e-2, betas=(0.9, 0.95))
    fused = FusedAdamW(model, lr=1e-9, weight_decay=5e-2, betas=(0.9, 0.95))
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, opt in (("torch_adamw", ref), ("fused_adamw_plus_repack", fused)):
        model.zero_grad(set_to_none=True)
        loss, _, _ = model(imgs, mask_ratio=0.75)
        loss.backward()
        assert all(p.grad is not None for p in model.parameters() if p.requires_grad and p is not model.mask_token)
        for _ in range(2):
            opt.step()
        runs = []
        for _ in range(3):             # best of three back-to-back batches: a one-off host event (observed: ~90 ms once in a
            torch.cuda.synchronize()   # while at Huge, allocator traffic of the preceding backward) must not price the step
            t0 = time.perf_counter()
            for _ in range(iters):
                opt.step()
                if opt is fused:
                    model._ensure_packed(stream)
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0) / iters * 1e3)
        res[name] = round(min(runs), 3)
    model.zero_grad(set_to_none=True)
    return res


def input_pipeline_ms(bands, N, iters=10):
    """Not part of the metric (inputs are resident when timing starts): SURVEY 8f row N2, one batch of N cubes
    assembled on the device from HBM-resident synthetic scenes (hsimae_cube_gather through hsimae_amd.data).
    HBM-bound byte kernel: algorithmic bytes = read + write 324*bands B per cube."""
    import numpy as np
    import torch
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


def _oracle_rate(bands, n_sample, lt, ll, cores, budget_s):
    """Median patches/s of the CPU oracle's fwd+bwd at `cores` threads: 2 warm-ups, then 5 timed steps.  The
    sample is halved until one step fits budget_s / 7, so the leg stays bounded on a slow host; returns (rate, steps, sample).
    budget_s <= 0: the probe form (1 warm-up + 1 step)."""
    import torch
    from oracle import hsimae_oracle as O
    torch.set_num_threads(cores)
    cfg = O.OracleConfig(bands=bands)
    state = O.init_state(cfg, seed=0)
    g = torch.Generator().manual_seed(1234)
    xa = torch.rand(n_sample, 1, bands, 9, 9, generator=g)
    n1a, n2a = torch.rand(n_sample, cfg.T, generator=g), torch.rand(n_sample, 9, generator=g)

    def one(n):
        t0 = time.perf_counter()
        O.forward_backward(state, cfg, xa[:n], n1a[:n].numpy(), n2a[:n].numpy(), lt, ll)
        return time.perf_counter() - t0

    n = n_sample
    warm = one(n)                                                   # warm-up 1 (also calibrates the sample)
    if budget_s <= 0:
        return n / one(n), 1, n
    while warm > budget_s / 7 and n > 16:
        n //= 2
        warm = one(n)
    one(n)                                                          # warm-up 2
    ts = sorted(one(n) for _ in range(5))
    return n / ts[2], 5, n


def _log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(bands):
    """The CPU oracle (a port of the reference's algorithm, validated against it) on the host cores, fp32, at the
    workload's shape (batch 256) with every core and with 32 threads (torch's CPU ops on these small shapes stop scaling
    past a few dozen threads) — the better one is `value` — plus config 1 (Base, 48 bands, batch 64).  About 20 s in all."""
    allc = os.cpu_count() or 1
    tried = {}
    # thread counts: 32 (where torch's CPU ops on these small shapes stop scaling) and every core only on hosts with up to
    # 64 of them — on the 256-thread GPU hosts the all-core setting never finished a batch-32 probe in 25 s (rounds 1-2),
    # so it is no longer attempted (that probe alone was 25 s of every bench run)
    cand = sorted({min(allc, 32), allc}) if allc <= 64 else [32]
    for cores in cand:
        # every setting is probed at batch 32 in a child process with a time limit first: with all cores of a 256-core host
        # torch's intra-op pool can take minutes per step on these small shapes; such a setting is recorded as timed out
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-probe", str(cores), "--probe-bands", str(bands)],
                               capture_output=True, text=True, timeout=25)
            probe = float(r.stdout.strip().splitlines()[-1])
        except (subprocess.TimeoutExpired, ValueError, IndexError):
            probe = 0.0
        _log(f"cpu probe {cores} threads: {probe:.1f} patches/s ({time.perf_counter() - t0:.1f} s)")
        if probe <= 0.0 or (tried and probe < 0.5 * max(v[0] for v in tried.values())):
            tried[cores] = (probe, 0, 32)
            continue
        rate, steps, nsamp = _oracle_rate(bands, 256, 3 if bands == 96 else 6, 9, cores, 18.0)
        tried[cores] = (rate, steps, nsamp)
        _log(f"cpu baseline {cores} threads: {rate:.1f} patches/s")
    best = max(tried, key=lambda c: tried[c][0])
    c1_rate, c1_steps, c1_n = _oracle_rate(48, 64, 2, 7, best, 6.0)
    return {"value": round(tried[best][0], 2), "unit": "patches/s", "cores": best, "kind": "port",
            "sample": f"oracle fwd+bwd fp32, HSIMAE-Base 9x9x{bands}, batch {tried[best][2]}, median of {tried[best][1]} steps after 2 warm-ups",
            "threads_tried": {str(c): (round(v[0], 2) if v[0] > 0 else "probe timed out (>25 s at batch 32)") for c, v in tried.items()},
            "host_cores": allc,
            "not_tried": (f"{allc} threads: torch's intra-op pool does not finish a batch-32 probe in 25 s on this host class "
                          "(measured in rounds 1-2)") if allc > 64 else None,
            "config1": {"value": round(c1_rate, 2), "unit": "patches/s", "cores": best,
                        "sample": f"oracle fwd+bwd fp32, HSIMAE-Base 9x9x48, batch {c1_n}, median of {c1_steps} steps after 2 warm-ups"}}


def pct(sorted_vals, q):
    if not sorted_vals:
        return None
    i = min(len(sorted_vals) - 1, max(0, int(round(q * (len(sorted_vals) - 1)))))
    return sorted_vals[i]


# --------------------------------------------------------------------------- one rank
def dry_run(args, world, rank):
    """Launcher / protocol check without a GPU (tests/test_bench_launcher.py): gloo ranks, barrier, MAX reduce, one line."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    for _ in range(args.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    verify = None
    if world > 1 and args.verify:
        # the --verify protocol on synthetic gradients: a bucketed reduction of a rank-dependent buffer against one plain collective.
        # HSIMAE_DRYRUN_FAULT (tests): "order" = rank 1 reports its collectives in another order, "grad" = rank 1's reduced buffer
        # differs in one element, "grid" = rank 1 drew another grid — each must come back as dp_consistent = false.
        from hsimae_amd.parallel import GradReducer, verify_step
        total = 200_000
        base = (torch.arange(total, dtype=torch.float32) % 977) + 1.0
        flat = base * (rank + 1) / world
        ref = flat.clone()
        red = GradReducer(bucket_bytes=64 << 10)
        ranges = [(o, min(7_000, total - o)) for o in range(0, total, 7_000)][::-1]      # back to front, as the backward reports
        red.make_callback(flat)
        for st, (o, ln) in enumerate(ranges):
            red._on_range(st, o, ln, None)
        red.finish()
        dist.all_reduce(ref)
        launched, grids = list(red.launched), [(3, 9), (9, 3)]
        fault = os.environ.get("HSIMAE_DRYRUN_FAULT", "")
        if rank == 1 and fault == "order":
            launched[0], launched[1] = launched[1], launched[0]
        if rank == 1 and fault == "grad":
            flat[total // 2] += 1.0
        if rank == 1 and fault == "grid":
            grids[1] = (3, 9)
        verify = verify_step(flat, ref, grids, launched, 0.0)
    if rank == 0:
        N = args.batch or 4
        line = {"metric": "HSI patches/sec (dry run)", "value": round(world * N * args.steps / float(t), 1),
                "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(float(t) / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "none", "data": "dry-run",
                "config": {"workload": "launcher dry run (no GPU work)", "parallelism": f"dp{world}"}}
        if verify is not None:
            line["dp_consistent"], line["dp_verify"] = verify["dp_consistent"], verify
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    # before anything initialises HIP (hsimae_amd/__init__.py does the same on import): see the comment there
    if not os.environ.get("HSIMAE_KEEP_HW_QUEUES"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (weak scaling); 0 = the model's default")
    ap.add_argument("--model", default="base", choices=sorted(MODELS))
    ap.add_argument("--precision", default=None, choices=["bf16", "fp8"],
                    help="GEMM operand type of the encoder linears (default: bf16; huge: fp8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the step timing (profiling runs)")
    ap.add_argument("--force-ddp", action="store_true", help="run the RCCL gradient reducer even with one rank (test)")
    ap.add_argument("--init-pg-only", action="store_true", help="experiment: create the RCCL process group but do not use it")
    ap.add_argument("--dry-run", action="store_true", help="launcher / protocol check on CPU with gloo (no GPU work)")
    ap.add_argument("--verify", dest="verify", action="store_true", default=None,
                    help="data-parallel self-check after the timed steps (default: on whenever the gradient reducer runs): one extra "
                         "step in deterministic mode, its bucketed + overlapped reduction compared with ONE plain all-reduce of the "
                         "same step's local gradients, hashes of the reduced buffer / grid sequence / collective order all-gathered; "
                         "rank 0 prints dp_consistent and the bucket list")
    ap.add_argument("--no-verify", dest="verify", action="store_false")
    ap.add_argument("--watchdog", type=float, default=420.0,
                    help="seconds the legs after the timed region may take before rank 0 prints the line without them (0 = off)")
    ap.add_argument("--cpu-probe", type=int, default=0, help=argparse.SUPPRESS)      # child of cpu_baseline(): threads to probe
    ap.add_argument("--probe-bands", type=int, default=96, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_probe:
        rate, _, _ = _oracle_rate(args.probe_bands, 32, 3 if args.probe_bands == 96 else 6, 9, args.cpu_probe, 0.0)
        print(rate)
        return

    if args.verify is None:
        args.verify = True                            # (only acts where a reducer exists: world > 1 or --force-ddp)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        sys.exit(dry_run(args, world, rank))

    import torch
    import torch.distributed as dist
    from hsimae_amd import HSIMAE, swiglu_hidden
    from hsimae_amd.build import kernel_source_hash

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_ddp = world > 1 or args.force_ddp
    if args.init_pg_only and not use_ddp:             # (scripts/exp_ddp_slow.sh: what creating the communicator alone costs the step)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        if os.environ.get("HSIMAE_EXP_PG_ALLREDUCE") == "1":
            t_ = torch.ones(8, device=dev); dist.all_reduce(t_); torch.cuda.synchronize()
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    bands, D, heads, N0 = MODELS[args.model]
    precision = args.precision or ("fp8" if args.model == "huge" else "bf16")
    effective_note = None
    torch.manual_seed(0)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # the module prints what the reference's constructor prints; stdout carries the JSON line only
        model = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=bands, b_patch_size=8, embed_dim=D, depth=12,
                       num_heads=heads, s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8,
                       norm_pix_loss=True, trunc_init=True).to(dev)
    if precision == "fp8":
        model.set_precision("fp8")
        # what the encoder linears really run in: below embed_dim 512 "fp8" keeps the bf16 kernels (hsimae_effective_precision),
        # and the line must then say bf16 and be priced against the bf16 peak
        from hsimae_amd import _lib
        if _lib.load().hsimae_effective_precision(C.byref(model._config())) != _lib.PREC_FP8:
            precision = "bf16"
            effective_note = "requested fp8; at this width the bf16 kernels run (hsimae_effective_precision)"
    if use_ddp:
        model.enable_data_parallel(force_collectives=args.force_ddp)
    random.seed(0)                                    # same (len_t, len_l) sequence on every rank
    torch.manual_seed(1234 + rank)
    N = args.batch or N0
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
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    grids = []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
        grids.append((model.len_t, model.len_l))
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
        lt0, ll0 = HSIMAE.grid_candidates(bands // 8, 9, 0.75)[0]
        fl, fl_enc = flops_per_sample(bands, D, 12, 9, 64, 8, lt0, ll0, h, hd, parts=True)
        peak = PEAK_FP8_TFLOPS if precision == "fp8" else PEAK_BF16_TFLOPS
        value = world * N * args.steps / dt
        step_tflops = value * fl / 1e12 / world
        ev = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        evs = sorted(ev)
        per_grid = {}
        for g, e in zip(grids, ev):
            per_grid.setdefault(f"{g[0]}x{g[1]}", []).append(e)
        traffic = profile_traffic(args.model if precision == "bf16" else f"{args.model}_{precision}")
        opdesc = ("bf16 MFMA operands" if precision == "bf16" else
                  "e4m3 MX-scaled MFMA operands in the encoder linears (bf16 elsewhere)") + " / fp32 accumulate + residual"
        out = {
            "metric": f"HSI patches/sec (9x9x{bands}, mask 75%) pretrain fwd+bwd", "value": round(value, 1),
            "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": precision, "data": "synthetic",
            "config": {"workload": f"HSIMAE-{args.model.capitalize()} pretrain fwd+bwd, 9x9x{bands} cubes, per-GPU batch {N}, "
                                   f"mask 0.75, {opdesc}",
                       "per_gpu_batch": N, "global_batch": N * world, "parallelism": f"dp{world}"},
            "per_gpu": round(value / world, 1), "loss": round(last_loss, 6),
            # what the loaded library was built from (hsimae_build_info): an ablation / instrumented build names its switches here (and the
            # loader refuses it unless HSIMAE_ALLOW_VARIANT=1); default_flags false = a tuning knob was overridden at build time
            "build": _build_info(),
            "gflop_per_patch": round(fl / 1e9, 4),
            "roofline": {"bound": "mfma", "achieved": round(step_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(step_tflops / peak, 4),
                         "traffic": traffic["hbm_bytes_per_step"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         # the PMC passes are a separate run (rocprofv3 --pmc cannot ride along with the timed steps): the file
                         # records the hash of the kernel sources it was measured on; stale = the kernels have changed since
                         "traffic_stale": (traffic.get("kernel_source_sha") != kernel_source_hash()) if traffic else None,
                         "what": "whole step, algorithmic fwd+bwd FLOPs per GPU (SURVEY 8d), wall-clock bracket"},
            "step_ms": {"median": round(pct(evs, 0.5), 3), "p10": round(pct(evs, 0.1), 3), "p90": round(pct(evs, 0.9), 3),
                        "per_grid": {k: {"n": len(v), "median": round(pct(sorted(v), 0.5), 3)} for k, v in per_grid.items()},
                        "clock": "HIP events between consecutive steps on the launch stream"},
        }
        # The legs below (reducer detached, self-check, replays, CPU baseline) run AFTER the timed region.  If one of them does not come
        # back — a collective that another rank never entered on a node this code has not run on yet — the measurement must not
        # be lost with it: after --watchdog seconds rank 0 prints the line as it stands and leaves.
        watchdog = _Watchdog(args.watchdog, out)
    comm = None
    if use_ddp:
        # what the gradient exchange costs this configuration: the same step with the reducer detached (no collectives,
        # no callbacks), timed on every rank over a few steps; exposed = step(ddp) - step(no reduce)
        red = model._reducer
        launched = list(red.launched)
        model._reducer = None
        k = max(3, min(10, args.steps))
        step(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tn = torch.tensor([(time.perf_counter() - t1) / k], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tn, op=dist.ReduceOp.MAX)
        model._reducer = red
        comm = {"buckets": len(launched), "bytes": int(sum(hi - lo for lo, hi in launched) * 4),
                "step_ms_no_reduce": round(float(tn.item()) * 1e3, 3),
                "exposed_ms": round(dt / args.steps * 1e3 - float(tn.item()) * 1e3, 3),
                "transport": "RCCL all-reduce(sum) per bucket on the reducer's launch stream, fp32, pre-scaled by 1/world"}

    verify = None
    if use_ddp and args.verify:
        # The first multi-rank run must DETECT a wrong bucket order or a collective that overtook its kernels, not only time the
        # step (no 8-GPU node has run this path yet).  One more step in deterministic mode (bit-reproducible local gradients):
        #   (1) with the reducer: bucketed all-reduces launched from inside the backward on the reducer's stream;
        #   (2) the SAME step (RNG streams restored) with the reducer detached, then ONE plain all-reduce of the whole flat
        #       buffer after a device synchronise (the step keeps folding 1 / world into dLoss, so SUM is the mean on both legs);
        # (1) must equal (2) up to the summation order inside the collective, on every rank, and every rank must hold
        # bit-identical reduced gradients, the same grid sequence and the same list of collectives (hsimae_amd/parallel.py).
        from hsimae_amd.parallel import verify_step
        red = model._reducer
        prev_det = model._deterministic
        model.deterministic = True
        rng = (random.getstate(), torch.get_rng_state(), torch.cuda.get_rng_state(dev))
        lv = float(step().item()); torch.cuda.synchronize()
        g_b, launched_v, grid_v = model._flat_grad.clone(), list(red.launched), (model.len_t, model.len_l)
        random.setstate(rng[0]); torch.set_rng_state(rng[1]); torch.cuda.set_rng_state(rng[2], dev)
        model._reducer = None
        # detached, but with the SAME 1 / world folded into the bf16 dL/dpred: for a world that is not a power of two a division
        # after the all-reduce would move every bf16 rounding of the backward (1e-3 relative against rtol 2e-6: ADVICE r05)
        model._world_override = world
        step(); torch.cuda.synchronize()
        g_ref = model._flat_grad.clone()
        model._world_override = None
        model._reducer = red
        model.deterministic = prev_det
        if world > 1:
            dist.all_reduce(g_ref)                    # SUM of pre-scaled gradients = the mean, as the bucketed path computes it
        verify = verify_step(g_b, g_ref, grids + [grid_v], launched_v, lv)
        del g_b, g_ref

    if rank == 0:
        if comm is not None:
            out["comm"] = comm
        if verify is not None:
            out["dp_consistent"], out["dp_verify"] = verify["dp_consistent"], verify
        if effective_note:
            out["config"]["precision_note"] = effective_note
        _log(f"step timing done: {dt / args.steps * 1e3:.3f} ms/step")
        if not args.no_extras:
            enc_ms = encoder_only_ms(model, imgs)
            _log(f"encoder-only {enc_ms:.3f} ms")
            enc_tf = N * fl_enc / (enc_ms * 1e-3) / 1e12
            out["encoder_mfma_frac"] = {"achieved": round(enc_tf, 1), "peak": peak, "unit": "TFLOP/s",
                                        "frac": round(enc_tf / peak, 4), "ms": round(enc_ms, 3),
                                        "what": "hsimae_encode + hsimae_encode_backward, algorithmic encoder FLOPs, HIP events"}
            # the decoder group (the lowest MFMA fraction of the step): 8 blocks + embed / assembly / pred, forward and backward
            dfw, dbw = decoder_only_ms(model, imgs)
            fl_dec = fl - fl_enc
            out["roofline_decoder"] = {
                "bound": "mfma", "what": "hsimae_decode / hsimae_decode_backward, algorithmic decoder FLOPs (fwd 1/3, bwd 2/3), HIP events",
                "fwd_ms": round(dfw, 3), "bwd_ms": round(dbw, 3), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "fwd_achieved": round(N * fl_dec / 3 / (dfw * 1e-3) / 1e12, 1), "bwd_achieved": round(N * fl_dec * 2 / 3 / (dbw * 1e-3) / 1e12, 1),
                "achieved": round(N * fl_dec / ((dfw + dbw) * 1e-3) / 1e12, 1),
                "frac": round(N * fl_dec / ((dfw + dbw) * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                "per_block_us": {"fwd": round(dfw * 1e3 / 8, 1), "bwd": round(dbw * 1e3 / 8, 1),
                                 "note": "group time / 8 blocks (includes the embed / assembly / pred kernels' ~5 %)"}}
            _log(f"decoder-only fwd {dfw:.3f} ms, bwd {dbw:.3f} ms")
            blk = decoder_block_replay(model, N, (bands // 8) * 9, PEAK_BF16_TFLOPS)
            if blk is not None:
                out["roofline_decoder_block"] = blk
            K_tok = lt0 * ll0
            kr = kernel_rooflines(model, N * K_tok, PEAK_BF16_TFLOPS)
            out["roofline_kernel"], out["roofline_kernel_hbm"] = kr[0]
            if len(kr) > 1:
                out["roofline_kernel2"], out["roofline_kernel2_hbm"] = kr[1]
            _log("kernel replays done")
            out["optimizer_step_ms"] = optimizer_step_ms(model, imgs)
            out["input_pipeline"] = input_pipeline_ms(bands, N)
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(bands)
        watchdog.finish()                  # prints the line (once: the timer cannot fire any more)
    if use_ddp:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    C.CDLL(None).fflush(None)


if __name__ == "__main__":
    main()
