"""Per-kernel parity on a real MI355X, every call through the C ABI (libhsimae_hip.so).

Floating-point kernels are compared with a torch fp32 reference of the same op evaluated on the SAME
bf16-rounded operands (so only accumulation order / output rounding differ); tolerances are written at each
assert.  Integer outputs (masking) are compared bit-exactly with the oracle and the reference's fixtures.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from hsimae_amd import _lib
from oracle import hsimae_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def stream():
    return torch.cuda.current_stream().cuda_stream


def rup(x, m):
    return (x + m - 1) // m * m


def pack(sources, N_img, K_img):
    """sources: list of (fp32 matrix [rows, cols] on device, transpose, n_off, k_off) -> packed bf16 image."""
    lib = _lib.load()
    img = torch.zeros(N_img * K_img, dtype=torch.bfloat16, device=DEV)
    descs = (_lib.PackDesc * len(sources))()
    keep = []
    for i, (w, tr, n_off, k_off) in enumerate(sources):
        w = w.contiguous().float()
        keep.append(w)
        descs[i] = _lib.PackDesc(src=w.data_ptr(), rows=w.shape[0], cols=w.shape[1], transpose=tr, n_off=n_off,
                                 k_off=k_off, KS=K_img // 32, dst=img.data_ptr())
    host = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).clone()
    table = host.to(DEV)
    _lib.check(lib.hsimae_pack_matrix(table.data_ptr(), len(sources), max(w.numel() for w in keep), stream()))
    torch.cuda.synchronize()
    return img


def bf(x):
    return x.to(torch.bfloat16).float()


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12))


def gemm(akind, epi, **kw):
    p = _lib.GemmParams()
    for k, v in kw.items():
        setattr(p, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    _lib.check(_lib.load().hsimae_gemm(C.byref(p), akind, epi, stream()), "hsimae_gemm")
    torch.cuda.synchronize()


# ----------------------------------------------------------------------------------------------- GEMM family
@pytest.mark.parametrize("M,N,K", [(300, 128, 128), (128, 384, 128), (77, 80, 64), (513, 128, 352), (200, 256, 768), (130, 64, 96)])
def test_gemm_bf16_in_f32_out_asymmetric(M, N, K):
    torch.manual_seed(0)
    A = (torch.randn(M, K, device=DEV) * 0.5).to(torch.bfloat16)
    W = torch.randn(N, K, device=DEV) * 0.1
    bias = torch.randn(N, device=DEV)
    img = pack([(W, 0, 0, 0)], rup(N, 16), K)
    out = torch.full((M, N), float("nan"), device=DEV)
    gemm(_lib.A_BF16, _lib.E_F32, A=A, lda=K, M=M, N=rup(N, 16), K=K, n_valid=N, W=img, bias=bias, out=out, ldo=N)
    ref = A.float() @ bf(W).t() + bias
    assert rel_err(out, ref) < 2e-5          # fp32 accumulation-order differences only


def test_gemm_identity_with_asymmetric_B_catches_transposed_layouts():
    K = N = 64
    M = 64
    A = torch.eye(M, K, device=DEV).to(torch.bfloat16)
    W = (torch.arange(N * K, device=DEV).reshape(N, K) % 251).float()          # W[n,k] != W[k,n]
    img = pack([(W, 0, 0, 0)], N, K)
    out = torch.zeros(M, N, device=DEV)
    gemm(_lib.A_BF16, _lib.E_F32, A=A, lda=K, M=M, N=N, K=K, n_valid=N, W=img, out=out, ldo=N)
    assert torch.equal(out, bf(W).t().contiguous())


def test_pack_transposed_and_offset_placement():
    torch.manual_seed(1)
    d = 64
    Wq, Wk, Wv = (torch.randn(d, d, device=DEV) for _ in range(3))
    M = 100
    # fused [3d, d] image
    img = pack([(Wq, 0, 0, 0), (Wk, 0, d, 0), (Wv, 0, 2 * d, 0)], 3 * d, d)
    A = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    out = torch.zeros(M, 3 * d, device=DEV)
    gemm(_lib.A_BF16, _lib.E_F32, A=A, lda=d, M=M, N=3 * d, K=d, n_valid=3 * d, W=img, out=out, ldo=3 * d)
    ref = A.float() @ bf(torch.cat([Wq, Wk, Wv], 0)).t()
    assert rel_err(out, ref) < 2e-5
    # transposed image for the data gradient: dU = dQKV @ [Wq;Wk;Wv]
    imgT = pack([(Wq, 1, 0, 0), (Wk, 1, 0, d), (Wv, 1, 0, 2 * d)], d, 3 * d)
    dqkv = torch.randn(M, 3 * d, device=DEV).to(torch.bfloat16)
    du = torch.zeros(M, d, device=DEV)
    gemm(_lib.A_BF16, _lib.E_F32, A=dqkv, lda=3 * d, M=M, N=d, K=3 * d, n_valid=d, W=imgT, out=du, ldo=d)
    ref = dqkv.float() @ bf(torch.cat([Wq, Wk, Wv], 0))
    assert rel_err(du, ref) < 2e-5


@pytest.mark.parametrize("M,d,N", [(300, 128, 384), (20001, 128, 384), (150, 64, 192), (260, 256, 768), (129, 512, 128), (90, 32, 96)])
def test_gemm_layernorm_prologue_bias_bf16_out(M, d, N):
    torch.manual_seed(2)
    x = torch.randn(M, d, device=DEV) * 2 + 0.3
    gamma, beta = 1 + 0.1 * torch.randn(d, device=DEV), 0.1 * torch.randn(d, device=DEV)
    W = torch.randn(N, d, device=DEV) * 0.1
    bias = torch.randn(N, device=DEV) * 0.1
    img = pack([(W, 0, 0, 0)], N, d)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    u = torch.zeros(M, d, dtype=torch.bfloat16, device=DEV)
    gemm(_lib.A_F32_LN, _lib.E_BF16, A=x, lda=d, M=M, N=N, K=d, n_valid=N, W=img, bias=bias, gamma=gamma, beta=beta,
         u_out=u, ldu=d, out=out, ldo=N)
    un = torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5)
    assert float((u.float() - un).abs().max()) <= 2 ** -8 * float(un.abs().max())      # one bf16 rounding
    ref = u.float() @ bf(W).t() + bias
    assert float((out.float() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("M,acc", [(300, 0), (129, 1), (1000, 0)])
def test_gemm_layernorm_backward_epilogue(M, acc):
    """E_LN_BWD: dx = dres (+ dx) + LayerNormBackward(dqkv @ [Wq;Wk;Wv]) with dgamma / dbeta, against autograd."""
    torch.manual_seed(12)
    d = 128
    W = torch.randn(3 * d, d, device=DEV) * 0.1
    imgT = pack([(W[:d], 1, 0, 0), (W[d:2 * d], 1, 0, d), (W[2 * d:], 1, 0, 2 * d)], d, 3 * d)
    dqkv = torch.randn(M, 3 * d, device=DEV).to(torch.bfloat16)
    x = (torch.randn(M, d, device=DEV) * 2 + 0.5).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(d, device=DEV)).requires_grad_(True)
    beta = torch.zeros(d, device=DEV, requires_grad=True)
    dres = torch.randn(M, d, device=DEV)
    du = dqkv.float() @ bf(W)
    torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5).backward(du)
    prev = torch.randn(M, d, device=DEV)
    dx = prev.clone()
    dg, dbt = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    gemm(_lib.A_BF16, _lib.E_LN_BWD, A=dqkv, lda=3 * d, M=M, N=d, K=3 * d, n_valid=d, W=imgT, out=dx, ldo=d,
         res=dres, ldr=d, lnx=x.detach(), gamma=gamma.detach(), dgamma=dg, dbeta=dbt, accumulate=acc)
    ref = x.grad + dres + (prev if acc else 0)
    assert rel_err(dx, ref) < 2e-5
    assert rel_err(dg, gamma.grad) < 1e-4 and rel_err(dbt, beta.grad) < 1e-4
    # in place over the residual gradient (how hsimae_backward calls it)
    g1 = dres.clone()
    dg.zero_(); dbt.zero_()
    gemm(_lib.A_BF16, _lib.E_LN_BWD, A=dqkv, lda=3 * d, M=M, N=d, K=3 * d, n_valid=d, W=imgT, out=g1, ldo=d,
         res=g1, ldr=d, lnx=x.detach(), gamma=gamma.detach(), dgamma=dg, dbeta=dbt, accumulate=0)
    assert rel_err(g1, x.grad + dres) < 2e-5


def test_gemm_residual_and_pos_epilogues():
    torch.manual_seed(3)
    M, d, K = 333, 128, 352
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    W = torch.randn(d, K, device=DEV) * 0.05
    bias = torch.randn(d, device=DEV)
    res, res2 = torch.randn(M, d, device=DEV), torch.randn(M, d, device=DEV)
    img = pack([(W, 0, 0, 0)], d, K)
    out = torch.zeros(M, d, device=DEV)
    gemm(_lib.A_BF16, _lib.E_RES_F32, A=A, lda=K, M=M, N=d, K=K, n_valid=d, W=img, bias=bias, res=res, res2=res2, ldr=d,
         out=out, ldo=d)
    ref = A.float() @ bf(W).t() + bias + res + res2
    assert rel_err(out, ref) < 2e-5
    pos = torch.randn(50, d, device=DEV)
    ids = torch.randint(0, 50, (M,), device=DEV, dtype=torch.int32)
    gemm(_lib.A_BF16, _lib.E_POS_F32, A=A, lda=K, M=M, N=d, K=K, n_valid=d, W=img, bias=bias, pos=pos, ids=ids, ldpos=d,
         out=out, ldo=d)
    ref = A.float() @ bf(W).t() + bias + pos[ids.long()]
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("d,h", [(128, 344), (64, 172), (256, 684), (32, 88)])
def test_gemm_swiglu_forward_and_backward_epilogues(d, h):
    torch.manual_seed(4)
    M, hp = 210, rup(h, 32)
    x = torch.randn(M, d, device=DEV)
    gamma, beta = 1 + 0.1 * torch.randn(d, device=DEV), 0.1 * torch.randn(d, device=DEV)
    W1, W3 = torch.randn(h, d, device=DEV) * 0.2, torch.randn(h, d, device=DEV) * 0.2
    b1, b3 = torch.randn(h, device=DEV) * 0.1, torch.randn(h, device=DEV) * 0.1
    i1, i3 = pack([(W1, 0, 0, 0)], hp, d), pack([(W3, 0, 0, 0)], hp, d)
    g = torch.full((M, hp), 7.0, dtype=torch.bfloat16, device=DEV)
    h13 = torch.full((M, 2 * hp), 7.0, dtype=torch.bfloat16, device=DEV)
    u = torch.zeros(M, d, dtype=torch.bfloat16, device=DEV)
    gemm(_lib.A_F32_LN, _lib.E_SWIGLU, A=x, lda=d, M=M, N=hp, K=d, n_valid=h, W=i1, W2=i3, bias=b1, bias2=b3,
         gamma=gamma, beta=beta, u_out=u, ldu=d, out=g, ldo=hp, h13=h13, ldh=2 * hp, hoff=hp)
    h1 = u.float() @ bf(W1).t() + b1
    h3 = u.float() @ bf(W3).t() + b3
    assert float((h13[:, :h].float() - h1).abs().max()) <= 2 ** -8 * float(h1.abs().max()) + 1e-6
    assert float((h13[:, hp:hp + h].float() - h3).abs().max()) <= 2 ** -8 * float(h3.abs().max()) + 1e-6
    assert float(g[:, h:].abs().max()) == 0 and float(h13[:, h:hp].abs().max()) == 0    # K-padding is zero
    gref = torch.nn.functional.silu(h13[:, :h].float()) * h13[:, hp:hp + h].float()
    assert float((g[:, :h].float() - gref).abs().max()) <= 2 ** -8 * float(gref.abs().max()) + 1e-6
    # backward epilogue: dg = dY @ W2 (W2 [d,h]); dh1 = dg*h3*silu'(h1); dh3 = dg*silu(h1)
    W2 = torch.randn(d, h, device=DEV) * 0.2
    i2T = pack([(W2, 1, 0, 0)], hp, d)
    dY = torch.randn(M, d, device=DEV)
    dh13 = torch.full((M, 2 * hp), 7.0, dtype=torch.bfloat16, device=DEV)
    gemm(_lib.A_F32, _lib.E_SWIGLU_BWD, A=dY, lda=d, M=M, N=hp, K=d, n_valid=hp, W=i2T, out=dh13, ldo=2 * hp,
         h13=h13, ldh=2 * hp, hoff=hp)
    dg = bf(dY) @ bf(W2)
    a1, a3 = h13[:, :h].float(), h13[:, hp:hp + h].float()
    s = torch.sigmoid(a1)
    r1 = dg * a3 * s * (1 + a1 * (1 - s))
    r3 = dg * a1 * s
    assert float((dh13[:, :h].float() - r1).abs().max()) <= 2 ** -7 * float(r1.abs().max())
    assert float((dh13[:, hp:hp + h].float() - r3).abs().max()) <= 2 ** -7 * float(r3.abs().max())
    assert float(dh13[:, h:hp].abs().max()) == 0


@pytest.mark.parametrize("M,drop", [(210, False), (4133, False), (333, True)])
def test_fused_encoder_mlp_half_forward_backward(M, drop):
    """hsimae_enc_mlp_fwd / hsimae_enc_mlp_bwd (one kernel each way, Base width) against a plain PyTorch fp32
    restatement of `x + rs * mlp(norm2(x))` (Models.py:305, 231-232) and its autograd, ragged row counts, optional
    DropPath row factors; also the bf16 weight-gradient operands the backward emits."""
    torch.manual_seed(6)
    d, h = 128, 344
    hp = rup(h, 32)
    lib = _lib.load()
    x1 = torch.randn(M, d, device=DEV)
    dy = torch.randn(M, d, device=DEV) * 0.1
    gamma, beta = 1 + 0.1 * torch.randn(d, device=DEV), 0.1 * torch.randn(d, device=DEV)
    W1, W3, W2 = (torch.randn(h, d, device=DEV) * 0.1, torch.randn(h, d, device=DEV) * 0.1, torch.randn(d, h, device=DEV) * 0.1)
    b1, b3, b2 = torch.randn(h, device=DEV) * 0.1, torch.randn(h, device=DEV) * 0.1, torch.randn(d, device=DEV) * 0.1
    rs_m = rs_a = None
    if drop:                                               # factors 0 or 1/keep, constant over runs of 9 rows
        rs_m = (torch.rand((M + 8) // 9, device=DEV) < 0.8).float().div(0.8).repeat_interleave(9)[:M].contiguous()
        rs_a = (torch.rand((M + 8) // 9, device=DEV) < 0.8).float().div(0.8).repeat_interleave(9)[:M].contiguous()
    i1, i3 = pack([(W1, 0, 0, 0)], hp, d), pack([(W3, 0, 0, 0)], hp, d)
    i2 = pack([(W2, 0, 0, 0)], d, hp)
    i2T = pack([(W2, 1, 0, 0)], hp, d)
    i13T = pack([(W1, 1, 0, 0), (W3, 1, 0, hp)], d, 2 * hp)
    b1p, b3p = torch.zeros(hp, device=DEV), torch.zeros(hp, device=DEV)
    b1p[:h], b3p[:h] = b1, b3
    w = _lib.MlpWeights(n2w=gamma.data_ptr(), n2b=beta.data_ptr(), w1b=b1p.data_ptr(), w3b=b3p.data_ptr(), w2b=b2.data_ptr(),
                        w1=i1.data_ptr(), w3=i3.data_ptr(), w2=i2.data_ptr(), w2T=i2T.data_ptr(), w13T=i13T.data_ptr(), hidden=h)
    x2 = torch.empty(M, d, device=DEV)
    _lib.check(lib.hsimae_enc_mlp_fwd(x1.data_ptr(), None, x2.data_ptr(), M, d, C.byref(w), _lib.ptr(rs_m), stream()))
    # reference, fp32 with the bf16-rounded weights the kernel multiplies with
    xr = x1.clone().requires_grad_(True)
    u = torch.nn.functional.layer_norm(xr, (d,), gamma, beta, 1e-5)
    h1, h3 = u @ bf(W1).t() + b1, u @ bf(W3).t() + b3
    gate = torch.nn.functional.silu(h1) * h3
    branch = gate @ bf(W2).t() + b2
    y = xr + (branch * rs_m[:, None] if drop else branch)
    torch.cuda.synchronize()
    assert rel_err(x2, y.detach()) < 6e-3                   # bf16 operands (u, gate), fp32 accumulation
    y.backward(dy)
    dx1 = torch.empty(M, d, device=DEV)
    u2, dyb, dx1b = (torch.empty(M, d, dtype=torch.bfloat16, device=DEV) for _ in range(3))
    dh13 = torch.empty(M, 2 * hp, dtype=torch.bfloat16, device=DEV)
    g = torch.empty(M, hp, dtype=torch.bfloat16, device=DEV)
    gw, gb = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    _lib.check(lib.hsimae_enc_mlp_bwd(x1.data_ptr(), dy.data_ptr(), dx1.data_ptr(), u2.data_ptr(), dh13.data_ptr(), g.data_ptr(),
                                      dyb.data_ptr(), dx1b.data_ptr(), M, d, C.byref(w), gw.data_ptr(), gb.data_ptr(),
                                      _lib.ptr(rs_m), _lib.ptr(rs_a), 0, stream()))
    torch.cuda.synchronize()
    assert rel_err(dx1, xr.grad) < 1.5e-2
    assert rel_err(u2.float(), u.detach()) < 2 ** -7
    assert rel_err(g[:, :h].float(), gate.detach()) < 1e-2 and float(g[:, h:].float().abs().max()) == 0
    dyr = dy * rs_m[:, None] if drop else dy
    assert rel_err(dyb.float(), dyr) < 2 ** -7                                       # the MLP branch's incoming gradient
    assert rel_err(dx1b.float(), xr.grad * rs_a[:, None] if drop else xr.grad) < 1.5e-2   # what the attention branch sees
    # dh1 | dh3: check through the weight gradients they produce (what they are for)
    dW1 = dh13[:, :h].float().t() @ u2.float()
    dW3 = dh13[:, hp:hp + h].float().t() @ u2.float()
    dgate = dyr @ bf(W2)
    s = torch.sigmoid(h1.detach())
    r1 = (dgate * h3.detach() * s * (1 + h1.detach() * (1 - s))).t() @ u.detach()
    r3 = (dgate * h1.detach() * s).t() @ u.detach()
    assert rel_err(dW1, r1) < 2e-2 and rel_err(dW3, r3) < 2e-2
    assert float(dh13[:, h:hp].float().abs().max()) == 0
    # second residual operand (the last spectral block adds the other axis stack's output)
    res2 = torch.randn(M, d, device=DEV)
    x2r = torch.full((M, d), 7.0, device=DEV)
    _lib.check(lib.hsimae_enc_mlp_fwd(x1.data_ptr(), res2.data_ptr(), x2r.data_ptr(), M, d, C.byref(w), _lib.ptr(rs_m), stream()))
    torch.cuda.synchronize()
    assert torch.equal(x2r, x2 + res2) or rel_err(x2r, y.detach() + res2) < 6e-3
    # LayerNorm-2 parameter gradients (accumulated with atomics)
    xh = (x1 - x1.mean(1, keepdim=True)) / torch.sqrt(x1.var(1, unbiased=False, keepdim=True) + 1e-5)
    dgate_u = (dgate * h3.detach() * s * (1 + h1.detach() * (1 - s))) @ bf(W1) + (dgate * h1.detach() * s) @ bf(W3)
    assert rel_err(gw, (dgate_u * xh).sum(0)) < 2e-2 and rel_err(gb, dgate_u.sum(0)) < 2e-2


@pytest.mark.parametrize("M,pad", [(96, 0), (4128, 48), (110592 // 8, 48), (4128, 16)])
def test_planar_weight_gradient_operands(M, pad):
    """Round 5: g / dh1 / dh3 of the fused MLP backward as 64-column planes [plane][R][64], R = M + pad rows per plane
    (hsimae_wgrad_task.dO_plane_rows / A_plane_rows; enc_mlp_bwd then writes row-contiguous blocks instead of 128-byte pieces
    at the row pitch).  (1) the planar buffers hold exactly the row-major values, plane by plane, and nothing else of the
    buffer is written; (2) the weight-gradient launch on planar operands returns the dW / db of the row-major launch; (3) a
    planar launch whose row count is not a multiple of 32, or whose planes are shorter than M, is refused."""
    torch.manual_seed(8)
    d, h = 128, 344
    hp, hp64 = rup(h, 32), rup(h, 64)
    P, R = hp64 // 64, M + pad
    lib = _lib.load()
    x1 = torch.randn(M, d, device=DEV)
    dy = torch.randn(M, d, device=DEV) * 0.1
    gamma, beta = 1 + 0.1 * torch.randn(d, device=DEV), 0.1 * torch.randn(d, device=DEV)
    W1, W3, W2 = (torch.randn(h, d, device=DEV) * 0.1, torch.randn(h, d, device=DEV) * 0.1, torch.randn(d, h, device=DEV) * 0.1)
    b2 = torch.randn(d, device=DEV) * 0.1
    b1p, b3p = torch.zeros(hp, device=DEV), torch.zeros(hp, device=DEV)
    b1p[:h], b3p[:h] = torch.randn(h, device=DEV) * 0.1, torch.randn(h, device=DEV) * 0.1
    imgs = [pack([(W1, 0, 0, 0)], hp, d), pack([(W3, 0, 0, 0)], hp, d), pack([(W2, 0, 0, 0)], d, hp), pack([(W2, 1, 0, 0)], hp, d),
            pack([(W1, 1, 0, 0), (W3, 1, 0, hp)], d, 2 * hp)]
    w = _lib.MlpWeights(n2w=gamma.data_ptr(), n2b=beta.data_ptr(), w1b=b1p.data_ptr(), w3b=b3p.data_ptr(), w2b=b2.data_ptr(),
                        w1=imgs[0].data_ptr(), w3=imgs[1].data_ptr(), w2=imgs[2].data_ptr(), w2T=imgs[3].data_ptr(),
                        w13T=imgs[4].data_ptr(), hidden=h)

    def run(plane_rows):
        dx1 = torch.empty(M, d, device=DEV)
        u2, dyb, dx1b = (torch.empty(M, d, dtype=torch.bfloat16, device=DEV) for _ in range(3))
        n13, ng = (2 * P * plane_rows * 64, P * plane_rows * 64) if plane_rows else (M * 2 * hp, M * hp)
        dh13 = torch.full((n13,), 7.0, dtype=torch.bfloat16, device=DEV)        # sentinel: unwritten elements stay 7
        g = torch.full((ng,), 7.0, dtype=torch.bfloat16, device=DEV)
        gw, gb = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
        _lib.check(lib.hsimae_enc_mlp_bwd(x1.data_ptr(), dy.data_ptr(), dx1.data_ptr(), u2.data_ptr(), dh13.data_ptr(), g.data_ptr(),
                                          dyb.data_ptr(), dx1b.data_ptr(), M, d, C.byref(w), gw.data_ptr(), gb.data_ptr(),
                                          None, None, plane_rows, stream()))
        torch.cuda.synchronize()
        return dx1, u2, dyb, dh13, g

    dx1_r, u2, dyb, dh13_r, g_r = run(0)
    dx1_p, _, _, dh13_p, g_p = run(R)
    assert torch.equal(dx1_r, dx1_p)
    g_r, dh13_r = g_r.view(M, hp), dh13_r.view(M, 2 * hp)
    gp, d1p, d3p = g_p.view(P, R, 64), dh13_p[:P * R * 64].view(P, R, 64), dh13_p[P * R * 64:].view(P, R, 64)
    for c in range(P):
        ncol = min(64, hp - 64 * c)
        assert torch.equal(gp[c, :M, :ncol], g_r[:, 64 * c:64 * c + ncol])
        assert torch.equal(d1p[c, :M, :ncol], dh13_r[:, 64 * c:64 * c + ncol])
        assert torch.equal(d3p[c, :M, :ncol], dh13_r[:, hp + 64 * c:hp + 64 * c + ncol])
        if ncol < 64:                                         # the columns past hp of the last plane are not written ...
            assert float((gp[c, :, ncol:].float() - 7.0).abs().max()) == 0
        if pad:                                               # ... nor are the rows past M of any plane
            assert float((gp[c, M:].float() - 7.0).abs().max()) == 0 and float((d3p[c, M:].float() - 7.0).abs().max()) == 0

    def wgrad(planar, Mrows=M, rows=R):
        wp = _lib.WgradParams()
        outs = []
        for i in range(3):
            N, K = (h, d) if i < 2 else (d, h)
            dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
            outs.append((dW, db))
            if i < 2:       # dW1 / dW3: dO = dh1 / dh3, A = u2
                if planar:
                    t = _lib.WgradTask(dO=dh13_p.data_ptr() + 2 * i * P * R * 64, dO_f32=0, ldo=hp64, A=u2.data_ptr(), lda=d, N=N, K=K,
                                       dW=dW.data_ptr(), ldw=K, db=db.data_ptr(), dO_plane_rows=rows)
                else:
                    t = _lib.WgradTask(dO=dh13_r.data_ptr() + 2 * i * hp, dO_f32=0, ldo=2 * hp, A=u2.data_ptr(), lda=d, N=N, K=K,
                                       dW=dW.data_ptr(), ldw=K, db=db.data_ptr())
            else:           # dW2: dO = dY (bf16), A = g
                t = _lib.WgradTask(dO=dyb.data_ptr(), dO_f32=0, ldo=d, A=(g_p if planar else g_r).data_ptr(), lda=hp64 if planar else hp,
                                   N=N, K=K, dW=dW.data_ptr(), ldw=K, db=db.data_ptr(), A_plane_rows=rows if planar else 0)
            wp.t[i] = t
        wp.ntasks, wp.M, wp.msplit = 3, Mrows, max(1, min(16, Mrows // 64))
        rc = lib.hsimae_wgrad(C.byref(wp), stream())
        torch.cuda.synchronize()
        return rc, outs

    rc_r, o_r = wgrad(False)
    rc_p, o_p = wgrad(True)
    assert rc_r == 0 and rc_p == 0
    for (a_, ab), (b_, bb) in zip(o_r, o_p):
        assert rel_err(b_, a_) < 1e-5 and rel_err(bb, ab) < 1e-5        # same products, fp32 atomics in another order
    assert wgrad(True, Mrows=M - 8)[0] == -1                            # planes need whole 32-row chunks
    assert wgrad(True, rows=M - 32)[0] == -1                            # ... and at least M rows each


# ----------------------------------------------------------------------------------------------- fused decoder Block
@pytest.mark.parametrize("N,Ts,split,slab", [(5, 108, 1, True), (5, 108, 0, False), (3, 54, 1, True), (4, 72, 1, True), (2, 90, 0, True),
                                             (7, 112, 1, False), (300, 108, 1, True), (3, 17, 1, True)])
def test_fused_decoder_block_forward_backward(N, Ts, split, slab):
    """hsimae_dec_block_fwd / _bwd (the decoder Block of hsimae_forward / hsimae_backward: attention half with q / k / v in
    registers + row-panel MLP half, or the one-kernel forward; two persistent backward kernels + the slab reduce) against a
    plain PyTorch fp32 Block (Models.py:303-306, 192-232) with the bf16-rounded weights, through autograd: x1, x2, O, dx and all
    18 parameter gradients.  Sequence lengths with key tiles that are partly / wholly padding, 17-token sequences, more samples
    than workgroups (300 > 256: a workgroup walks two samples)."""
    torch.manual_seed(11 + Ts)
    d, heads, hd, h = 64, 8, 8, 172
    hp = rup(h, 32)
    lib = _lib.load()
    M = N * Ts
    x = torch.randn(M, d, device=DEV)
    dy = torch.randn(M, d, device=DEV) * 0.1
    P = {k: torch.randn(d, d, device=DEV) * 0.15 for k in ("q", "k", "v", "p")}
    P.update(w1=torch.randn(h, d, device=DEV) * 0.12, w3=torch.randn(h, d, device=DEV) * 0.12, w2=torch.randn(d, h, device=DEV) * 0.1)
    B = {k: torch.randn(n, device=DEV) * 0.1 for k, n in (("q", d), ("k", d), ("v", d), ("p", d), ("w1", h), ("w3", h), ("w2", d))}
    n1w, n1b, n2w, n2b = (1 + 0.1 * torch.randn(d, device=DEV), 0.1 * torch.randn(d, device=DEV),
                          1 + 0.1 * torch.randn(d, device=DEV), 0.1 * torch.randn(d, device=DEV))
    iqkv = pack([(P["q"], 0, 0, 0), (P["k"], 0, d, 0), (P["v"], 0, 2 * d, 0)], 3 * d, d)
    ip = pack([(P["p"], 0, 0, 0)], d, d)
    i1, i3 = pack([(P["w1"], 0, 0, 0)], hp, d), pack([(P["w3"], 0, 0, 0)], hp, d)
    i2, i2T = pack([(P["w2"], 0, 0, 0)], d, hp), pack([(P["w2"], 1, 0, 0)], hp, d)
    bqkv = torch.cat([B["q"], B["k"], B["v"]]).contiguous()
    W = _lib.DecBlockWeights(n1w=n1w.data_ptr(), n1b=n1b.data_ptr(), bqkv=bqkv.data_ptr(), pb=B["p"].data_ptr(), n2w=n2w.data_ptr(),
                             n2b=n2b.data_ptr(), w1b=B["w1"].data_ptr(), w3b=B["w3"].data_ptr(), w2b=B["w2"].data_ptr(),
                             qkv=iqkv.data_ptr(), p=ip.data_ptr(), w1=i1.data_ptr(), w3=i3.data_ptr(), w2=i2.data_ptr(), w2T=i2T.data_ptr(),
                             qf=P["q"].data_ptr(), kf=P["k"].data_ptr(), vf=P["v"].data_ptr(), pf=P["p"].data_ptr(),
                             w1f=P["w1"].data_ptr(), w3f=P["w3"].data_ptr(), hidden=h)
    x1, x2 = torch.full((M, d), float("nan"), device=DEV), torch.full((M, d), float("nan"), device=DEV)
    o = torch.zeros(M, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(M, heads, device=DEV)
    _lib.check(lib.hsimae_dec_block_fwd(C.byref(W), x.data_ptr(), x1.data_ptr(), x2.data_ptr(), o.data_ptr(), lse.data_ptr(),
                                        N, Ts, split, stream()), "hsimae_dec_block_fwd")
    # ---- reference: fp32, the weights the kernels multiply with (bf16-rounded), torch autograd
    leaves = {k: bf(v).clone().requires_grad_(True) for k, v in P.items()}
    bl = {k: v.clone().requires_grad_(True) for k, v in B.items()}
    ln = {k: v.clone().requires_grad_(True) for k, v in (("n1w", n1w), ("n1b", n1b), ("n2w", n2w), ("n2b", n2b))}
    xr = x.clone().requires_grad_(True)
    u = torch.nn.functional.layer_norm(xr, (d,), ln["n1w"], ln["n1b"], 1e-5)
    q, k, v = (u @ leaves[n].t() + bl[n] for n in ("q", "k", "v"))
    sh = lambda z: z.reshape(N, Ts, heads, hd).permute(0, 2, 1, 3)
    att = ((sh(q) @ sh(k).transpose(-1, -2)) * hd ** -0.5).softmax(-1) @ sh(v)
    oref = att.permute(0, 2, 1, 3).reshape(M, d)
    y1 = xr + oref @ leaves["p"].t() + bl["p"]
    u2 = torch.nn.functional.layer_norm(y1, (d,), ln["n2w"], ln["n2b"], 1e-5)
    y2 = y1 + (torch.nn.functional.silu(u2 @ leaves["w1"].t() + bl["w1"]) * (u2 @ leaves["w3"].t() + bl["w3"])) @ leaves["w2"].t() + bl["w2"]
    torch.cuda.synchronize()
    assert rel_err(o.float(), oref.detach()) < 1.5e-2            # bf16 q / k / v / P operands
    assert rel_err(x1, y1.detach()) < 1e-2 and rel_err(x2, y2.detach()) < 1e-2
    y2.backward(dy)
    # ---- backward through the C ABI
    names = ("n1w", "n1b", "qw", "qb", "kw", "kb", "vw", "vb", "pw", "pb", "n2w", "n2b", "w1w", "w1b", "w2w", "w2b", "w3w", "w3b")
    shapes = dict(n1w=(d,), n1b=(d,), qw=(d, d), qb=(d,), kw=(d, d), kb=(d,), vw=(d, d), vb=(d,), pw=(d, d), pb=(d,), n2w=(d,), n2b=(d,),
                  w1w=(h, d), w1b=(h,), w2w=(d, h), w2b=(d,), w3w=(h, d), w3b=(h,))
    G_ = {n: torch.zeros(shapes[n], device=DEV) for n in names}
    Gs = _lib.DecBlockGrads(**{n: G_[n].data_ptr() for n in names})
    dx1, dx = torch.empty(M, d, device=DEV), torch.full((M, d), float("nan"), device=DEV)
    sl = torch.empty(lib.hsimae_dec_block_slab_floats(), device=DEV) if slab else None
    _lib.check(lib.hsimae_dec_block_bwd(C.byref(W), C.byref(Gs), x.data_ptr(), x1.data_ptr(), dy.data_ptr(), dx1.data_ptr(), dx.data_ptr(),
                                        o.data_ptr(), lse.data_ptr(), N, Ts, _lib.ptr(sl), stream()), "hsimae_dec_block_bwd")
    torch.cuda.synchronize()
    assert rel_err(dx, xr.grad) < 2e-2
    ref = dict(n1w=ln["n1w"].grad, n1b=ln["n1b"].grad, qw=leaves["q"].grad, qb=bl["q"].grad, kw=leaves["k"].grad, kb=bl["k"].grad,
               vw=leaves["v"].grad, vb=bl["v"].grad, pw=leaves["p"].grad, pb=bl["p"].grad, n2w=ln["n2w"].grad, n2b=ln["n2b"].grad,
               w1w=leaves["w1"].grad, w1b=bl["w1"].grad, w2w=leaves["w2"].grad, w2b=bl["w2"].grad, w3w=leaves["w3"].grad, w3b=bl["w3"].grad)
    for n in names:
        if n == "kb":                                         # exactly zero in exact arithmetic (softmax shift invariance)
            assert float(G_[n].abs().max()) <= 3e-2 * float(ref["qb"].abs().max()) + 1e-6
            continue
        e = float((G_[n].double() - ref[n].double()).pow(2).mean().sqrt() / ref[n].double().pow(2).mean().sqrt().clamp_min(1e-30))
        assert e < 2.5e-2, (n, e)


# ----------------------------------------------------------------------------------------------- attention
def attn_reference(qkv, d, heads, Ts, mode, len_l):
    """fp32 masked attention over [nsamples, Ts] tokens; qkv fp32 leaf [rows, 3d] (values already bf16-exact)."""
    rows = qkv.shape[0]
    n = rows // Ts
    hd = d // heads
    q, k, v = (qkv[:, i * d:(i + 1) * d].reshape(n, Ts, heads, hd).permute(0, 2, 1, 3) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * hd ** -0.5
    idx = torch.arange(Ts, device=qkv.device)
    cls = idx // len_l if mode == 1 else idx % len_l if mode == 2 else torch.zeros_like(idx)
    allow = cls[:, None] == cls[None, :]
    s = s.masked_fill(~allow, float("-inf"))
    p = s.softmax(-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(rows, d)


@pytest.mark.parametrize("d,heads,Ts,mode,len_l", [
    (128, 8, 27, 0, 9), (128, 8, 27, 1, 9), (128, 8, 27, 2, 9), (128, 8, 27, 1, 3), (128, 8, 27, 2, 3),
    (128, 8, 14, 1, 7), (128, 8, 14, 2, 7), (64, 8, 108, 0, 9), (64, 8, 54, 0, 9), (64, 8, 216, 0, 9),
    (32, 2, 18, 1, 9), (32, 2, 18, 2, 6), (32, 4, 36, 0, 9), (256, 16, 27, 2, 9), (512, 32, 54, 0, 9)])
def test_attention_forward_backward(d, heads, Ts, mode, len_l):
    torch.manual_seed(5)
    lib = _lib.load()
    n = 5
    rows = n * Ts
    qkv = (torch.randn(rows, 3 * d, device=DEV) * 1.5).to(torch.bfloat16)
    o = torch.zeros(rows, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(rows, heads, device=DEV)
    p = _lib.AttnParams(qkv=qkv.data_ptr(), ld=3 * d, d=d, heads=heads, hd=d // heads, Ts=Ts, nsamples=n, mode=mode,
                        len_l=len_l, o=o.data_ptr(), ldo=d, lse=lse.data_ptr())
    _lib.check(lib.hsimae_attn_fwd(C.byref(p), stream()), "attn_fwd")
    torch.cuda.synchronize()
    leaf = qkv.float().requires_grad_(True)
    ref = attn_reference(leaf, d, heads, Ts, mode, len_l)
    # bf16 probabilities + bf16 output rounding: 2^-8 relative each
    assert float((o.float() - ref).abs().max()) <= 3 * 2 ** -8 * float(ref.abs().max())
    do = (torch.randn(rows, d, device=DEV)).to(torch.bfloat16)
    ref.backward(do.float())
    dqkv = torch.full((rows, 3 * d), float("nan"), dtype=torch.bfloat16, device=DEV)
    p.dout, p.lddo, p.dqkv = do.data_ptr(), d, dqkv.data_ptr()
    _lib.check(lib.hsimae_attn_bwd(C.byref(p), stream()), "attn_bwd")
    torch.cuda.synchronize()
    assert torch.isfinite(dqkv.float()).all()
    for i, name in enumerate("qkv"):
        got, want = dqkv[:, i * d:(i + 1) * d].float(), leaf.grad[:, i * d:(i + 1) * d]
        err = float((got - want).abs().max()) / float(want.abs().max())
        rms = float((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
        assert err < 3e-2 and rms < 8e-3, (name, err, rms)      # bf16 P / dS operands (2^-8) through two products


# ----------------------------------------------------------------------------------------------- wgrad
@pytest.mark.parametrize("M,N,K,f32", [(1000, 128, 128, 0), (777, 344, 128, 0), (500, 128, 344, 1), (300, 72, 64, 0), (640, 128, 72, 1),
                                       # N, K >= 256: the 256 x 256 tiles (8 waves, one workgroup per CU), whole and ragged tiles
                                       (2000, 256, 256, 0), (5003, 704, 256, 0), (1500, 256, 696, 0), (9000, 1376, 512, 0),
                                       (700, 512, 1376, 0), (1500, 256, 256, 1)])
def test_wgrad_and_bias_grad(M, N, K, f32):
    torch.manual_seed(6)
    lib = _lib.load()
    ldo, lda = rup(N, 32) + 32, rup(K, 32)
    dO = torch.zeros(M, ldo, device=DEV)
    dO[:, :N] = torch.randn(M, N, device=DEV)
    dO_dev = dO if f32 else dO.to(torch.bfloat16)
    A = torch.zeros(M, lda, dtype=torch.bfloat16, device=DEV)
    A[:, :K] = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    dW = torch.zeros(N, K, device=DEV)
    db = torch.zeros(N, device=DEV)
    wp = _lib.WgradParams()
    wp.t[0] = _lib.WgradTask(dO=dO_dev.data_ptr(), dO_f32=f32, ldo=ldo, A=A.data_ptr(), lda=lda, N=N, K=K,
                             dW=dW.data_ptr(), ldw=K, db=db.data_ptr())
    wp.ntasks, wp.M, wp.msplit = 1, M, 3
    _lib.check(lib.hsimae_wgrad(C.byref(wp), stream()), "wgrad")
    torch.cuda.synchronize()
    dOr = bf(dO[:, :N])
    ref = dOr.t() @ A[:, :K].float()
    assert rel_err(dW, ref) < 5e-5
    assert rel_err(db, dOr.sum(0)) < 5e-5


def test_wgrad_wide_block_batch():
    """One wide transformer block's seven linears in one launch, laid out as hsimae_backward passes them at d = 256: q | k | v as
    column ranges of one dqkv slab, w1 | w3 of one dh13 slab; gradients accumulate onto what the buffers hold."""
    torch.manual_seed(16)
    lib = _lib.load()
    M, d, h = 3100, 256, 680
    hp = rup(h, 32)
    dqkv = torch.randn(M, 3 * d, device=DEV).to(torch.bfloat16)
    u = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    g1b = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    o = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    dh13 = torch.zeros(M, 2 * hp, device=DEV, dtype=torch.bfloat16)
    dh13[:, :h] = torch.randn(M, h, device=DEV).to(torch.bfloat16)
    dh13[:, hp:hp + h] = torch.randn(M, h, device=DEV).to(torch.bfloat16)
    u2 = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    g0b = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    gt = torch.zeros(M, hp, device=DEV, dtype=torch.bfloat16)
    gt[:, :h] = torch.randn(M, h, device=DEV).to(torch.bfloat16)
    specs = [(dqkv, 0, 3 * d, u, d, d, d), (dqkv, d, 3 * d, u, d, d, d), (dqkv, 2 * d, 3 * d, u, d, d, d), (g1b, 0, d, o, d, d, d),
             (dh13, 0, 2 * hp, u2, d, h, d), (dh13, hp, 2 * hp, u2, d, h, d), (g0b, 0, d, gt, hp, d, h)]
    wp = _lib.WgradParams()
    outs = []
    for i, (dO, off, ldo, A, lda, N, K) in enumerate(specs):
        dW = torch.full((N, K), 0.5, device=DEV)
        db = torch.full((N,), -1.0, device=DEV)
        outs.append((dW, db))
        wp.t[i] = _lib.WgradTask(dO=dO.data_ptr() + 2 * off, dO_f32=0, ldo=ldo, A=A.data_ptr(), lda=lda, N=N, K=K,
                                 dW=dW.data_ptr(), ldw=K, db=db.data_ptr())
    wp.ntasks, wp.M, wp.msplit = len(specs), M, 8
    _lib.check(lib.hsimae_wgrad(C.byref(wp), stream()), "wgrad")
    torch.cuda.synchronize()
    for (dO, off, ldo, A, lda, N, K), (dW, db) in zip(specs, outs):
        dOr = dO[:, off:off + N].float()
        assert rel_err(dW - 0.5, dOr.t() @ A[:, :K].float()) < 5e-5
        assert rel_err(db + 1.0, dOr.sum(0)) < 5e-5


@pytest.mark.parametrize("M,msplit", [(3100, 8), (110592 // 8, 59), (45, 3)])
def test_wgrad_narrow_block_batch(M, msplit):
    """One d = 128 encoder block's seven linears in one launch as hsimae_backward passes them: q | k | v as three 128-column slices of
    one dqkv slab over the same A, the projection (128 x 128), w1 | w3 as slices of one dh13 slab (352 x 128 each), w2 (128 x 352);
    ragged row counts, a row count smaller than the row split, accumulation onto what the buffers hold."""
    torch.manual_seed(17)
    lib = _lib.load()
    d, h = 128, 344
    hp = rup(h, 32)
    dqkv = torch.randn(M, 3 * d, device=DEV).to(torch.bfloat16)
    u = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    g1b = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    o = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    dh13 = torch.zeros(M, 2 * hp, device=DEV, dtype=torch.bfloat16)
    dh13[:, :h] = torch.randn(M, h, device=DEV).to(torch.bfloat16)
    dh13[:, hp:hp + h] = torch.randn(M, h, device=DEV).to(torch.bfloat16)
    u2 = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    g0b = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    gt = torch.zeros(M, hp, device=DEV, dtype=torch.bfloat16)
    gt[:, :h] = torch.randn(M, h, device=DEV).to(torch.bfloat16)
    specs = [(dqkv, 0, 3 * d, u, d, d, d), (dqkv, d, 3 * d, u, d, d, d), (dqkv, 2 * d, 3 * d, u, d, d, d), (g1b, 0, d, o, d, d, d),
             (dh13, 0, 2 * hp, u2, d, h, d), (dh13, hp, 2 * hp, u2, d, h, d), (g0b, 0, d, gt, hp, d, h)]
    wp = _lib.WgradParams()
    outs = []
    for i, (dO, off, ldo, A, lda, N, K) in enumerate(specs):
        dW = torch.full((N, K), 0.5, device=DEV)
        db = torch.full((N,), -1.0, device=DEV)
        outs.append((dW, db))
        wp.t[i] = _lib.WgradTask(dO=dO.data_ptr() + 2 * off, dO_f32=0, ldo=ldo, A=A.data_ptr(), lda=lda, N=N, K=K,
                                 dW=dW.data_ptr(), ldw=K, db=db.data_ptr())
    wp.ntasks, wp.M, wp.msplit = len(specs), M, msplit
    _lib.check(lib.hsimae_wgrad(C.byref(wp), stream()), "wgrad")
    torch.cuda.synchronize()
    for (dO, off, ldo, A, lda, N, K), (dW, db) in zip(specs, outs):
        dOr = dO[:, off:off + N].float()
        assert rel_err(dW - 0.5, dOr.t() @ A[:, :K].float()) < 5e-5, (N, K)
        assert rel_err(db + 1.0, dOr.sum(0)) < 5e-5, (N, K)


# ----------------------------------------------------------------------------------------------- LayerNorm bwd / fwd
@pytest.mark.parametrize("M,d,acc", [(300, 128, 0), (129, 64, 1), (70, 256, 0), (65, 512, 0), (50, 32, 1)])
def test_layernorm_backward(M, d, acc):
    torch.manual_seed(7)
    lib = _lib.load()
    x = (torch.randn(M, d, device=DEV) * 2 + 0.5).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(d, device=DEV)).requires_grad_(True)
    beta = torch.zeros(d, device=DEV, requires_grad=True)
    du, dres = torch.randn(M, d, device=DEV), torch.randn(M, d, device=DEV)
    y = torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5)
    y.backward(du)
    prev = torch.randn(M, d, device=DEV)
    dx = prev.clone()
    dg, dbt = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    p = _lib.LnBwdParams(du=du.data_ptr(), x=x.data_ptr(), stats=None, gamma=gamma.data_ptr(), dres=dres.data_ptr(),
                         dx=dx.data_ptr(), accumulate=acc, dgamma=dg.data_ptr(), dbeta=dbt.data_ptr(), M=M, d=d)
    _lib.check(lib.hsimae_ln_bwd(C.byref(p), stream()), "ln_bwd")
    torch.cuda.synchronize()
    ref = x.grad + dres + (prev if acc else 0)
    assert rel_err(dx, ref) < 1e-5
    assert rel_err(dg, gamma.grad) < 1e-4 and rel_err(dbt, beta.grad) < 1e-4      # atomic summation order
    out = torch.zeros(M, d, device=DEV)
    _lib.check(lib.hsimae_ln_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), M, d, stream()))
    torch.cuda.synchronize()
    assert rel_err(out, y.detach()) < 1e-5


# ----------------------------------------------------------------------------------------------- masking (bit-exact)
def run_mask(n1, n2, lt, ll):
    lib = _lib.load()
    N, T = n1.shape
    L = n2.shape[1]
    keep = torch.zeros(N, lt * ll, dtype=torch.int32, device=DEV)
    rest = torch.zeros(N, T * L, dtype=torch.int32, device=DEV)
    mask = torch.zeros(N, T * L, device=DEV)
    a, b = n1.to(DEV), n2.to(DEV)
    p = _lib.MaskParams(noise1=a.data_ptr(), noise2=b.data_ptr(), N=N, T=T, L=L, len_t=lt, len_l=ll,
                        ids_keep=keep.data_ptr(), ids_restore=rest.data_ptr(), mask=mask.data_ptr())
    _lib.check(lib.hsimae_mask_from_noise(C.byref(p), stream()), "mask")
    torch.cuda.synchronize()
    return keep.cpu().numpy(), rest.cpu().numpy(), mask.cpu().numpy()


def test_masking_bit_exact_vs_reference_fixtures():
    z = np.load(os.path.join(G, "masking.npz"))
    meta = json.load(open(os.path.join(G, "masking.json")))
    for c in meta["cases"]:
        k = c["key"]
        keep, rest, mask = run_mask(torch.from_numpy(z[k + "_n1"]), torch.from_numpy(z[k + "_n2"]), c["len_t"], c["len_l"])
        assert np.array_equal(keep, z[k + "_keep"].astype(np.int32))
        assert np.array_equal(rest, z[k + "_restore"].astype(np.int32))
        assert np.array_equal(mask, z[k + "_mask"].astype(np.float32))


def test_masking_full_size_ties_and_properties():
    torch.manual_seed(8)
    N, T, L, lt, ll = 4096, 12, 9, 3, 9
    n1, n2 = torch.rand(N, T), torch.rand(N, L)
    n1[:64, 3] = n1[:64, 7]                       # exact cross-group ties: lower index wins
    n2[:64] = 0.25
    keep, rest, mask = run_mask(n1, n2, lt, ll)
    k2, r2, m2 = O.mask_from_noise(n1.numpy(), n2.numpy(), lt, ll)
    assert np.array_equal(keep, k2) and np.array_equal(rest, r2) and np.array_equal(mask, m2)
    assert (np.sort(rest, 1) == np.arange(T * L)).all()                   # a permutation per sample
    assert (mask.sum(1) == T * L - lt * ll).all()
    assert (np.diff(keep, axis=1) > 0).all()
    assert np.array_equal(np.take_along_axis(rest, keep, 1), np.tile(np.arange(lt * ll), (N, 1)))
    # edge: N = 1, and keep-everything grids
    k3, r3, m3 = run_mask(torch.rand(1, 4), torch.rand(1, 9), 4, 9)
    assert k3.tolist() == [list(range(36))] and m3.sum() == 0


# ----------------------------------------------------------------------------------------------- patch gather / assemble / loss
@pytest.mark.parametrize("strided", [False, True])
def test_patch_gather_contiguous_and_band_fastest_layouts(strided):
    torch.manual_seed(9)
    lib = _lib.load()
    N, B, lt, ll = 7, 48, 2, 7
    cfg = O.OracleConfig(bands=B)
    x = torch.rand(N, 1, B, 9, 9)
    keep, _, _ = O.mask_from_noise(torch.rand(N, 6).numpy(), torch.rand(N, 9).numpy(), lt, ll)
    K = lt * ll
    xd = x.to(DEV)
    if strided:     # HSIdataset4PT layout (Model_Pretraining.py:49-50)
        xd = xd[:, 0].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).unsqueeze(1)
        assert not xd.is_contiguous()
    ids = torch.from_numpy(keep.astype(np.int32)).to(DEV)
    out = torch.full((N * K, 96), 5.0, dtype=torch.bfloat16, device=DEV)
    p = _lib.PatchParams(x=xd.data_ptr(), sn=xd.stride(0), sb=xd.stride(2), sh=xd.stride(3), sw=xd.stride(4), N=N, T=B // 8,
                         K=K, ids_keep=ids.data_ptr(), out=out.data_ptr(), pos_ids=None)
    _lib.check(lib.hsimae_patch_gather(C.byref(p), stream()), "patch_gather")
    torch.cuda.synchronize()
    ref = torch.gather(O.patchify(x, cfg), 1, torch.from_numpy(keep).unsqueeze(-1).expand(-1, -1, 72)).reshape(N * K, 72)
    assert torch.equal(out[:, :72].cpu().float(), bf(ref))
    assert float(out[:, 72:].abs().max()) == 0


def test_decoder_assembly_forward_backward():
    torch.manual_seed(10)
    lib = _lib.load()
    N, T, lt, ll, Dd = 6, 6, 2, 7, 64
    TL, K = T * 9, lt * ll
    _, rest, _ = O.mask_from_noise(torch.rand(N, T).numpy(), torch.rand(N, 9).numpy(), lt, ll)
    rest_t = torch.from_numpy(rest).to(DEV)
    y = torch.randn(N, K, Dd, device=DEV, requires_grad=True)
    pos = torch.randn(TL, Dd, device=DEV)
    yall = torch.cat([y, y.mean(1, keepdim=True).expand(N, TL - K, Dd)], 1)
    ref = torch.gather(yall, 1, rest_t.unsqueeze(-1).expand(-1, -1, Dd)) + pos
    ids = rest_t.int().contiguous()
    yfull = torch.zeros(N, TL, Dd, device=DEV)
    p = _lib.AssembleParams(y=y.data_ptr(), N=N, K=K, TL=TL, Dd=Dd, ids_restore=ids.data_ptr(), pos=pos.data_ptr(),
                            yfull=yfull.data_ptr())
    _lib.check(lib.hsimae_assemble_fwd(C.byref(p), stream()), "assemble_fwd")
    torch.cuda.synchronize()
    assert rel_err(yfull, ref.detach()) < 1e-6
    dyf = torch.randn(N, TL, Dd, device=DEV)
    ref.backward(dyf)
    dy = torch.zeros(N, K, Dd, dtype=torch.bfloat16, device=DEV)
    p.dyfull, p.dy = dyf.data_ptr(), dy.data_ptr()
    _lib.check(lib.hsimae_assemble_bwd(C.byref(p), stream()), "assemble_bwd")
    torch.cuda.synchronize()
    assert float((dy.float() - y.grad).abs().max()) <= 2 ** -8 * float(y.grad.abs().max())


@pytest.mark.parametrize("norm_pix", [1, 0])
def test_loss_dpred_and_recons(norm_pix):
    torch.manual_seed(11)
    lib = _lib.load()
    N, B, lt, ll = 5, 48, 2, 7
    cfg = O.OracleConfig(bands=B, norm_pix_loss=bool(norm_pix))
    T, TL, K = B // 8, B // 8 * 9, lt * ll
    x = torch.rand(N, 1, B, 9, 9)
    _, _, mask = O.mask_from_noise(torch.rand(N, T).numpy(), torch.rand(N, 9).numpy(), lt, ll)
    pred = torch.randn(N, TL, 72, requires_grad=True)
    tgt = O.patchify(x, cfg)
    if norm_pix:
        mean, std = tgt.mean(-1, keepdim=True), (tgt.var(-1, keepdim=True) + 1e-6) ** 0.5
        tgt = (tgt - mean) / std
    mk = torch.from_numpy(mask)
    loss_ref = ((((pred - tgt) ** 2).mean(-1)) * mk).sum() / mk.sum()
    loss_ref.backward()
    xd, pd, md = x.to(DEV), pred.detach().to(DEV), mk.to(DEV)
    nparts = lib.hsimae_loss_partials(N, T)
    partial = torch.zeros(nparts, device=DEV)
    loss = torch.zeros((), device=DEV)
    dpred = torch.full((N * TL, 96), 3.0, dtype=torch.bfloat16, device=DEV)
    pimg, mimg = torch.zeros(N, 1, B, 9, 9, device=DEV), torch.zeros(N, 1, B, 9, 9, device=DEV)
    sm = float(mk.sum())
    gs = 0.5
    p = _lib.LossParams(x=xd.data_ptr(), sn=xd.stride(0), sb=xd.stride(2), sh=xd.stride(3), sw=xd.stride(4), N=N, T=T,
                        pred=pd.data_ptr(), mask=md.data_ptr(), norm_pix=norm_pix, inv_scale=gs / (72 * sm),
                        partial=partial.data_ptr(), loss=loss.data_ptr(), sum_mask=sm, dpred=dpred.data_ptr(),
                        pred_img=pimg.data_ptr(), mask_img=mimg.data_ptr())
    _lib.check(lib.hsimae_loss(C.byref(p), stream()), "loss")
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) <= 2e-6 * abs(loss_ref.item())
    want = (pred.grad * gs).reshape(N * TL, 72)
    assert float((dpred[:, :72].float().cpu() - want).abs().max()) <= 2 ** -8 * float(want.abs().max())
    assert float(dpred[:, 72:].abs().max()) == 0
    p2 = pred.detach() * std + mean if norm_pix else pred.detach()
    assert rel_err(pimg.cpu(), O.unpatchify(p2, cfg)) < 1e-6
    assert torch.equal(mimg.cpu(), O.unpatchify(mk.unsqueeze(2).repeat(1, 1, 72), cfg))
