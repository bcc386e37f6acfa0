"""Times hsimae_enc_mlp_bwd at the decoder's shape (d = 64, hidden 172, M = 442,368) with and without the weight-gradient
operand outputs (GPU box): python3 scripts/micro/mlp_bwd_datapath.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsimae_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda"
for d, h, M in ((64, 172, 442368), (128, 344, 110592)):
    hp = (h + 31) // 32 * 32
    f32, bf = dict(dtype=torch.float32, device=dev), dict(dtype=torch.bfloat16, device=dev)
    x1, dy = torch.randn(M, d, **f32), torch.randn(M, d, **f32) * 1e-3
    dx1 = torch.empty(M, d, **f32)
    u2, dyb, dx1b = (torch.empty(M, d, **bf) for _ in range(3))
    dh13, g = torch.empty(M, 2 * hp, **bf), torch.empty(M, hp, **bf)
    w1, w3, w2T = (torch.randn(hp * d, **bf) * 0.05 for _ in range(3))
    w2, w13T = torch.randn(d * hp, **bf) * 0.05, torch.randn(d * 2 * hp, **bf) * 0.05
    n2w, n2b, b2 = torch.ones(d, **f32), torch.zeros(d, **f32), torch.zeros(d, **f32)
    b1, b3 = torch.zeros(hp, **f32), torch.zeros(hp, **f32)
    gw, gb = torch.zeros(d, **f32), torch.zeros(d, **f32)
    w = _lib.MlpWeights(n2w=n2w.data_ptr(), n2b=n2b.data_ptr(), w1b=b1.data_ptr(), w3b=b3.data_ptr(), w2b=b2.data_ptr(),
                        w1=w1.data_ptr(), w3=w3.data_ptr(), w2=w2.data_ptr(), w2T=w2T.data_ptr(), w13T=w13T.data_ptr(), hidden=h)
    s = torch.cuda.current_stream().cuda_stream
    for name, ops in (("all operands", (u2, dh13, g, dyb, dx1b)), ("no dh13/g", (u2, None, None, dyb, dx1b)),
                      ("data path only", (None, None, None, None, None))):
        p = [None if t is None else t.data_ptr() for t in ops]
        def launch():
            _lib.check(lib.hsimae_enc_mlp_bwd(x1.data_ptr(), dy.data_ptr(), dx1.data_ptr(), p[0], p[1], p[2], p[3], p[4], M, d,
                                              C.byref(w), gw.data_ptr(), gb.data_ptr(), None, None, s), "mlp_bwd")
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            launch()
        e1.record()
        torch.cuda.synchronize()
        print(f"d={d} h={h} M={M}: {name:16s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
