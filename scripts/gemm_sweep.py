#!/usr/bin/env python3
"""Tile sweep of the row-panel GEMM at the wide layers' shapes (GPU box): python scripts/gemm_sweep.py [large|huge]
Every (a_kind, epilogue, M, N, K) of one encoder block, each tiling (bm, kc) timed with HIP events, interleaved with a
cache-evicting fill so every launch starts from HBM."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsimae_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "large"
D, hp, M = (256, 704, 110592) if which == "large" else (512, 1376, 55296)
s = torch.cuda.current_stream().cuda_stream
f32 = dict(dtype=torch.float32, device=dev)
bf = dict(dtype=torch.bfloat16, device=dev)
evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def run(name, akind, epi, N, K, **kw):
    res = []
    for bm, kc in ((128, 128), (64, 128), (128, 256), (64, 256)):
        if akind == _lib.A_F32_LN and kc == 256:
            continue
        p = _lib.GemmParams()
        for k, v in kw.items():
            setattr(p, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        p.M, p.N, p.K = M, N, K
        ts = []
        for it in range(6):
            evict.fill_(it)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.hsimae_gemm_tiled(C.byref(p), akind, epi, bm, kc, s)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0, (name, rc)
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[1:])
        res.append(f"bm{bm}/kc{kc} {ts[len(ts) // 2]:7.1f}")
    print(f"{name:34s} N={N:5d} K={K:5d}  " + "   ".join(res), flush=True)


x = torch.randn(M, D, **f32)
ub = torch.randn(M, D, **bf)
g1 = torch.ones(D, **f32); b1 = torch.zeros(D, **f32)
W = torch.randn(4 * hp * D, **bf) * 0.02          # any content is a valid packed image
bias = torch.zeros(4 * hp, **f32)
out_b = torch.empty(M, 3 * D, **bf)
out_f = torch.empty(M, D, **f32)
u = torch.empty(M, D, **bf)
h13 = torch.empty(M, 2 * hp, **bf); g = torch.empty(M, hp, **bf)
dq = torch.randn(M, 3 * D, **bf)
dh13 = torch.randn(M, 2 * hp, **bf)
gb = torch.randn(M, hp, **bf)
A, E = _lib, _lib
run("LN1 + q|k|v", A.A_F32_LN, E.E_BF16, 3 * D, D, A=x, lda=D, n_valid=3 * D, W=W, bias=bias, gamma=g1, beta=b1, u_out=u, ldu=D, out=out_b, ldo=3 * D)
run("proj + residual", A.A_BF16, E.E_RES_F32, D, D, A=ub, lda=D, n_valid=D, W=W, bias=bias, res=x, ldr=D, out=out_f, ldo=D)
run("LN2 + w1|w3 + gate", A.A_F32_LN, E.E_SWIGLU, hp, D, A=x, lda=D, n_valid=hp - 8, W=W, W2=W, bias=bias, bias2=bias, gamma=g1, beta=b1,
    u_out=u, ldu=D, out=g, ldo=hp, h13=h13, ldh=2 * hp, hoff=hp)
run("w2 + residual", A.A_BF16, E.E_RES_F32, D, hp, A=gb, lda=hp, n_valid=D, W=W, bias=bias, res=x, ldr=D, out=out_f, ldo=D)
run("dg = dY W2 (gate bwd)", A.A_F32, E.E_SWIGLU_BWD, hp, D, A=x, lda=D, n_valid=hp, W=W, out=dh13, ldo=2 * hp, h13=h13, ldh=2 * hp, hoff=hp)
run("du2 = dh13 W13", A.A_BF16, E.E_F32, D, 2 * hp, A=dh13, lda=2 * hp, n_valid=D, W=W, out=out_f, ldo=D)
run("dO = dx1 Wp", A.A_BF16, E.E_BF16, D, D, A=ub, lda=D, n_valid=D, W=W, out=u, ldo=D)
run("du = dqkv Wqkv", A.A_BF16, E.E_F32, D, 3 * D, A=dq, lda=3 * D, n_valid=D, W=W, out=out_f, ldo=D)
