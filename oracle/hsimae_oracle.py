"""CPU oracle for the HSIMAE masked-autoencoder pretraining path.

TEST INFRASTRUCTURE ONLY.  This file is a from-scratch restatement (written from
the math in SURVEY.md Appendix A) of what the reference computes on its
pretraining hot path.  It exists so that the HIP kernels can be checked on a GPU
box where the reference cannot travel.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it; the product package
`hsimae_amd` never does.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the reference
(`/root/reference/Models.py`) in the build container, replays its RNG streams,
and commits the reference's own outputs as fixtures under `tests/golden/`;
`tests/test_oracle_golden.py` checks every function below against them
(ids bit-exact, fp32 loss to 1e-6 rel, activations/grads to 1e-5).  The fine-tuning additions (`encode_unmasked`,
`dualvit_classify`, `drop_rates`, `draw_drop_factors`, `dualvit_train_step`) are pinned the same way by
`tests/golden/make_golden_dualvit.py` (eval forward) and `make_golden_dualvit_train.py` (one training step of the
reference DualViT with DropPath: recorded draws, losses, logits, all gradients), checked in `tests/test_dualvit_cpu.py`.

Each function cites the reference lines it restates (paths relative to the
reference checkout).  Integer/index work is numpy; floating point is torch CPU
in a caller-chosen dtype (fp32 = what the reference runs, fp64 = tighter check).
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- config
@dataclass(frozen=True)
class OracleConfig:
    """Shape parameters of one HSIMAE instance (Models.py:312-332 ctor arguments)."""
    img_size: int = 9
    patch_size: int = 3
    bands: int = 96
    b_patch_size: int = 8
    embed_dim: int = 128
    depth: int = 12
    num_heads: int = 8
    s_depth: int = 9
    decoder_embed_dim: int = 64
    decoder_depth: int = 8
    decoder_num_heads: int = 8
    mlp_ratio: float = 4.0
    norm_pix_loss: bool = True

    @property
    def T(self):  # spectral groups, Models.py:144
        return self.bands // self.b_patch_size

    @property
    def grid(self):  # Models.py:143
        return self.img_size // self.patch_size

    @property
    def L(self):
        return self.grid * self.grid

    @property
    def patch_dim(self):  # Models.py:422
        return self.b_patch_size * self.patch_size ** 2


def swiglu_hidden(dim: int, mlp_ratio: float) -> int:
    """Models.py:225 as wired by Models.py:300-301 (multiple_of = mlp_ratio)."""
    hidden_dim = int(dim * mlp_ratio)
    multiple_of = mlp_ratio
    return int(multiple_of * ((2 * hidden_dim // 3 + multiple_of - 1) // multiple_of))


# --------------------------------------------------------------------------- pos embed
def _sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """Models.py:86-101."""
    omega = np.arange(embed_dim // 2, dtype=np.float32)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_pos_embed_3d(embed_dim: int, t_size: int, grid_size: int) -> np.ndarray:
    """Fixed 3-D sin-cos table [t_size*grid^2, embed_dim] (Models.py:11-47).

    First D/2 channels: 1-D table over the spectral index; last D/2: 2-D table over
    the grid, whose first half encodes the *w* coordinate (meshgrid "w goes first",
    Models.py:19) and second half the *h* coordinate.
    """
    assert embed_dim % 4 == 0
    ds = embed_dim // 2
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)
    emb_a = _sincos_1d(ds // 2, grid[0])
    emb_b = _sincos_1d(ds // 2, grid[1])
    spatial = np.concatenate([emb_a, emb_b], axis=1)               # [L, D/2]
    temporal = _sincos_1d(embed_dim // 2, np.arange(t_size, dtype=np.float32))  # [T, D/2]
    L = grid_size * grid_size
    te = np.repeat(temporal[:, None, :], L, axis=1)
    sp = np.repeat(spatial[None, :, :], t_size, axis=0)
    return np.concatenate([te, sp], axis=-1).reshape(-1, embed_dim).astype(np.float32)


# --------------------------------------------------------------------------- masking
def grid_candidates(T: int, L: int, mask_ratio: float):
    """All (len_t, len_l) minimising |(1-r)*T*L - t*l|, in the reference's order.

    Models.py:484-489.  The distance is evaluated the way the reference does:
    python float `len_keep` minus int64 product, promoted by torch to fp32.
    """
    pairs = list(itertools.product(range(2, T + 1), range(2, L + 1)))
    len_keep = (1 - mask_ratio) * T * L
    lens = torch.tensor([a * b for a, b in pairs])
    diff = abs(len_keep - lens)
    ind = torch.where(diff == torch.min(diff))[0].tolist()
    return [pairs[i] for i in ind]


def choose_grid(T: int, L: int, mask_ratio: float, py_random) -> tuple[int, int]:
    """Models.py:490-493: one `random.sample(range(n), 1)` draw, even when n == 1."""
    cands = grid_candidates(T, L, mask_ratio)
    j = py_random.sample(range(len(cands)), 1)[0]
    return cands[j]


def mask_from_noise(noise_1: np.ndarray, noise_2: np.ndarray, len_t: int, len_l: int):
    """Closed form of Models.py:495-535.

    noise_1 [N,T], noise_2 [N,L] fp32.  Keeps the len_t spectral groups with the
    smallest noise_1 and the len_l positions with the smallest noise_2 (ties: lower
    index wins).  Returns ids_keep [N,K] int64 ascending, ids_restore [N,T*L] int64,
    mask [N,T*L] fp32 (0 keep / 1 remove).
    """
    noise_1 = np.asarray(noise_1, dtype=np.float32)
    noise_2 = np.asarray(noise_2, dtype=np.float32)
    N, T = noise_1.shape
    L = noise_2.shape[1]
    TL = T * L
    keep_t = np.zeros((N, T), dtype=bool)
    keep_l = np.zeros((N, L), dtype=bool)
    o1 = np.argsort(noise_1, axis=1, kind="stable")[:, :len_t]
    o2 = np.argsort(noise_2, axis=1, kind="stable")[:, :len_l]
    np.put_along_axis(keep_t, o1, True, axis=1)
    np.put_along_axis(keep_l, o2, True, axis=1)
    # class 0: both kept; 1: exactly one kept; 2: neither (mask_1 + mask_2, Models.py:520)
    cls = (~keep_t)[:, :, None].astype(np.int64) + (~keep_l)[:, None, :].astype(np.int64)
    cls = cls.reshape(N, TL)
    key = cls * TL + np.arange(TL, dtype=np.int64)[None, :]       # all keys distinct
    ids_shuffle = np.argsort(key, axis=1, kind="stable")
    ids_restore = np.argsort(ids_shuffle, axis=1, kind="stable").astype(np.int64)
    K = len_t * len_l
    ids_keep = ids_shuffle[:, :K].astype(np.int64)
    mask = (cls != 0).astype(np.float32)
    return ids_keep, ids_restore, mask


def mask_from_noise_literal(noise_1: torch.Tensor, noise_2: torch.Tensor, len_t: int, len_l: int):
    """Step-by-step argsort route of Models.py:498-534 (used to pin the closed form)."""
    N, T = noise_1.shape
    L = noise_2.shape[1]
    mask_1 = torch.ones(N, T * L)
    mask_2 = torch.ones(N, T * L)
    n1 = noise_1.repeat_interleave(L, 1)
    ids_restore = torch.argsort(torch.argsort(n1, dim=1, stable=True), dim=1, stable=True)
    mask_1[:, : len_t * L] = 0
    mask_1 = torch.gather(mask_1, 1, ids_restore)
    n2 = noise_2.repeat(1, T)
    ids_restore = torch.argsort(torch.argsort(n2, dim=1, stable=True), dim=1, stable=True)
    mask_2[:, : len_l * T] = 0
    mask_2 = torch.gather(mask_2, 1, ids_restore)
    mask_all = mask_1 + mask_2 + torch.linspace(0, 0.5, T * L).unsqueeze(0).repeat(N, 1)
    ids_shuffle = torch.argsort(mask_all, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    ids_keep = ids_shuffle[:, : len_t * len_l]
    mask = torch.ones(N, T * L)
    mask[:, : len_t * len_l] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return ids_keep.numpy(), ids_restore.numpy(), mask.numpy()


# --------------------------------------------------------------------------- patch maps
def patchify(imgs: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
    """[N,1,B,H,W] -> [N, T*L, u*p*p]; feature order (u,p,q) (Models.py:461-473)."""
    N = imgs.shape[0]
    p, u, g, T = cfg.patch_size, cfg.b_patch_size, cfg.grid, cfg.T
    x = imgs.reshape(N, T, u, g, p, g, p)
    x = x.permute(0, 1, 3, 5, 2, 4, 6)
    return x.reshape(N, T * g * g, u * p * p)


def unpatchify(x: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
    """Inverse of `patchify` (Models.py:475-482)."""
    N = x.shape[0]
    p, u, g, T = cfg.patch_size, cfg.b_patch_size, cfg.grid, cfg.T
    x = x.reshape(N, T, g, g, u, p, p)
    x = x.permute(0, 1, 4, 2, 5, 3, 6)
    return x.reshape(N, 1, cfg.bands, cfg.img_size, cfg.img_size)


# --------------------------------------------------------------------------- blocks
# Operand rounding (test infrastructure for tests/test_gpu_boundary.py::test_loss_error_is_the_bf16_operand_rounding):
# `with operands_bf16():` makes the forward below round every MATRIX-PRODUCT OPERAND to bf16 at exactly the places where the HIP
# path does (LayerNorm outputs, q | k | v, the unnormalised softmax numerators exp(s - max), the attention
# output, the SwiGLU gate product, the patch values, all weight matrices) and keep everything else — accumulation, residual
# stream, LayerNorm statistics, softmax sums, biases, loss — in the working dtype.  It is NOT what the reference computes; it
# separates the error the bf16 operands make inherent from anything else the kernels might add.
_ROUND = None


class operands_bf16:
    def __enter__(self):
        global _ROUND
        self._prev, _ROUND = _ROUND, (lambda t: t.to(torch.bfloat16).to(t.dtype))
        return self

    def __exit__(self, *exc):
        global _ROUND
        _ROUND = self._prev
        return False


def _r(t):
    return t if _ROUND is None else _ROUND(t)


# MX e4m3 operands (test infrastructure for tests/test_gpu_fp8.py: the fp8 analogue of operands_bf16).
# `with operands_mx8():` additionally runs every linear of the ENCODER blocks — q | k | v, proj, w1 | w3, w2, and their data
# gradients — on operands quantised as the kernels quantise them under precision = FP8 (csrc/gemm.hip stage_store, csrc/pack.hip
# pack8): OCP e4m3, one e8m0 scale per 32 consecutive elements of the CONTRACTION axis, scale = 2^(floor(log2 amax) - 8), scaled
# values clamped to +-448, round to nearest even; fp32 accumulation.  Weight gradients keep bf16 operands, the decoder, patch
# embedding, attention core and loss are as under operands_bf16.  Like operands_bf16 it is NOT the reference: it says how much of
# the fp8 path's distance to the reference the operand format makes inherent.
_MX = False


def mx_e4m3(t: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """Quantise-dequantise `t` in MX e4m3 blocks of 32 along `dim` (the contraction axis of the product it enters)."""
    x = t.movedim(dim, -1)
    K = x.shape[-1]
    pad = (-K) % 32
    if pad:
        x = F.pad(x, (0, pad))
    xb = x.reshape(*x.shape[:-1], -1, 32)
    am = xb.abs().amax(-1, keepdim=True)
    e = torch.frexp(am)[1] - 1                                  # floor(log2 amax)  (amax = m * 2^e', m in [0.5, 1))
    eb = (e + 127 - 8).clamp(1, 254)                            # the kernels' biased e8m0 byte
    scale = torch.ldexp(torch.ones_like(am), eb - 127)
    q = (xb / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(t.dtype) * scale
    q = q.reshape(*x.shape)
    if pad:
        q = q[..., :K]
    return q.movedim(-1, dim)


class _MXLinear(torch.autograd.Function):
    """y = mx(x) mx(W)^T + b;  dx = mx(dy) mx'(W) with the blocks of W along its OUTPUT axis (the transposed image the data
    gradient multiplies with);  dW = bf16(dy)^T bf16(x), db = sum dy (the weight-gradient kernels keep bf16 operands)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_b = b is not None
        y = mx_e4m3(x, -1) @ mx_e4m3(W, -1).t()
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        bf = lambda t: t.to(torch.bfloat16).to(t.dtype)      # noqa: E731
        dx = mx_e4m3(dy, -1) @ mx_e4m3(W, 0)
        d2, x2 = bf(dy).reshape(-1, dy.shape[-1]), bf(x).reshape(-1, x.shape[-1])
        return dx, d2.t() @ x2, (dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_b else None)


class operands_mx8(operands_bf16):
    def __enter__(self):
        global _MX
        super().__enter__()
        self._prev_mx, _MX = _MX, True
        return self

    def __exit__(self, *exc):
        global _MX
        _MX = self._prev_mx
        return super().__exit__(*exc)


def _lin(x, W, b, pre):
    """One linear of a Block: x is the operand BEFORE any rounding (the MX path quantises the fp32 LayerNorm output directly)."""
    if _MX and not pre.startswith("decoder"):
        return _MXLinear.apply(x, W, b)
    return F.linear(_r(x), _r(W), b)


def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def attention(x, P, pre, heads):
    """Models.py:192-219: separate q/k/v linears, softmax(q k^T * hd^-0.5) v, proj."""
    Bs, S, C = x.shape
    hd = C // heads

    def lin(name):
        return _r(_lin(x, P[f"{pre}.{name}.weight"], P.get(f"{pre}.{name}.bias"), pre))

    q = lin("q").reshape(Bs, S, heads, hd).permute(0, 2, 1, 3)
    k = lin("k").reshape(Bs, S, heads, hd).permute(0, 2, 1, 3)
    v = lin("v").reshape(Bs, S, heads, hd).permute(0, 2, 1, 3)
    attn = (q @ k.transpose(-2, -1)) * hd ** -0.5
    if _ROUND is None:
        attn = attn.softmax(dim=-1)
        o = (attn @ v).transpose(1, 2).reshape(Bs, S, C)
    else:       # the kernels multiply the bf16-rounded numerators exp(s - max) with v and divide by the fp32 sum afterwards
        e = torch.exp(attn - attn.amax(dim=-1, keepdim=True))
        o = ((_r(e) @ v) / e.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(Bs, S, C)
    return _lin(_r(o), P[f"{pre}.proj.weight"], P[f"{pre}.proj.bias"], pre)


def swiglu(x, P, pre):
    """Models.py:231-232."""
    h1 = _lin(x, P[f"{pre}.w1.weight"], P[f"{pre}.w1.bias"], pre)
    h3 = _lin(x, P[f"{pre}.w3.weight"], P[f"{pre}.w3.bias"], pre)
    return _lin(_r(F.silu(h1) * h3), P[f"{pre}.w2.weight"], P[f"{pre}.w2.bias"], pre)


def block(x, P, pre, heads, drop=None):
    """Models.py:303-306.  drop_path is Identity in HSIMAE (Models.py:298); DualViT in training mode multiplies each
    residual branch by a per-sequence factor (0 or 1/keep_prob, Models.py:235-250): `drop` = (attn, mlp) vectors of
    length x.shape[0], either may be None."""
    a = attention(layer_norm(x, P[f"{pre}.norm1.weight"], P[f"{pre}.norm1.bias"]), P, f"{pre}.attn", heads)
    if drop is not None and drop[0] is not None:
        a = a * drop[0].to(a.dtype).reshape(-1, 1, 1)
    x = x + a
    m = swiglu(layer_norm(x, P[f"{pre}.norm2.weight"], P[f"{pre}.norm2.bias"]), P, f"{pre}.mlp")
    if drop is not None and drop[1] is not None:
        m = m * drop[1].to(m.dtype).reshape(-1, 1, 1)
    return x + m


# --------------------------------------------------------------------------- full forward
def forward(P: dict, cfg: OracleConfig, imgs: torch.Tensor, noise_1, noise_2, len_t: int, len_l: int,
            taps: dict | None = None, drops: list | None = None):
    """Full pretraining forward, Models.py:627-634, given the replayed noise.

    drops (DualViT training only): per encoder block, in execution order blocks_1[0..], blocks_2[0..], blocks[0..],
    a pair (attn, mlp) of per-sequence DropPath factors (see `block`); None = no stochastic depth.

    P: name -> tensor (reference state_dict layout), all of one float dtype.
    Returns (loss, pred_img, mask_img); optional `taps` dict collects stage outputs.
    """
    dt = imgs.dtype
    N = imgs.shape[0]
    T, L, D = cfg.T, cfg.L, cfg.embed_dim
    TL = T * L
    K = len_t * len_l
    tap = (lambda k, v: taps.__setitem__(k, v)) if taps is not None else (lambda k, v: None)

    # 1. tokenise == Conv3d(k=s=(u,p,p)) (Models.py:151-160)
    Pm = patchify(imgs, cfg)                                           # [N,TL,72]
    Wpe = P["patch_embed.proj.weight"].reshape(D, -1)
    X0 = _r(Pm) @ _r(Wpe).t() + P["patch_embed.proj.bias"]
    tap("patch_embed", X0)

    # 2-4. masking (Models.py:495-535)
    ids_keep, ids_restore, mask = mask_from_noise(np.asarray(noise_1), np.asarray(noise_2), len_t, len_l)
    ids_keep_t = torch.from_numpy(ids_keep)
    ids_restore_t = torch.from_numpy(ids_restore)
    mask_t = torch.from_numpy(mask).to(dt)
    tap("ids_keep", ids_keep_t); tap("ids_restore", ids_restore_t); tap("mask", mask_t)

    # 5. gather kept tokens + pos embed (Models.py:528, 548-550)
    idx = ids_keep_t.unsqueeze(-1).expand(-1, -1, D)
    X = torch.gather(X0, 1, idx) + torch.gather(P["pos_embed"].expand(N, -1, -1), 1, idx)
    tap("enc_in", X)

    # 7. axis stacks (Models.py:552-564)
    if cfg.s_depth > 0:
        x1 = X.reshape(N, len_t, len_l, D).reshape(N * len_t, len_l, D)
        x2 = X.reshape(N, len_t, len_l, D).permute(0, 2, 1, 3).reshape(N * len_l, len_t, D)
        for i in range(cfg.s_depth):
            x1 = block(x1, P, f"blocks_1.{i}", cfg.num_heads, drops[i] if drops else None)
        for i in range(cfg.s_depth):
            x2 = block(x2, P, f"blocks_2.{i}", cfg.num_heads, drops[cfg.s_depth + i] if drops else None)
        x1 = x1.reshape(N, K, D)
        x2 = x2.reshape(N, len_l, len_t, D).permute(0, 2, 1, 3).reshape(N, K, D)
        tap("x1", x1); tap("x2", x2)
        X = x1 + x2
    # 8. fusion (Models.py:566-570)
    if cfg.s_depth < 12:
        for i in range(cfg.depth - cfg.s_depth):
            X = block(X, P, f"blocks.{i}", cfg.num_heads,
                      drops[(2 * cfg.s_depth if cfg.s_depth > 0 else 0) + i] if drops else None)
    tap("fused", X)
    latent = layer_norm(X, P["norm.weight"], P["norm.bias"])
    tap("latent", latent)

    # 9. decoder input (Models.py:579-592)
    Y = F.linear(_r(latent), _r(P["decoder_embed.weight"]), P["decoder_embed.bias"])
    Dd = Y.shape[-1]
    m = Y.mean(1, keepdim=True)
    Yall = torch.cat([Y, m.expand(N, TL - K, Dd)], dim=1)
    Yfull = torch.gather(Yall, 1, ids_restore_t.unsqueeze(-1).expand(-1, -1, Dd)) + P["decoder_pos_embed"]
    tap("dec_in", Yfull)

    # 10. decoder (Models.py:595-600)
    Z = Yfull
    for i in range(cfg.decoder_depth):
        Z = block(Z, P, f"decoder_blocks.{i}", cfg.decoder_num_heads)
    tap("dec_out", Z)
    Z = layer_norm(Z, P["decoder_norm.weight"], P["decoder_norm.bias"])
    pred = F.linear(_r(Z), _r(P["decoder_pred.weight"]), P["decoder_pred.bias"])
    tap("pred", pred)

    # 11. loss (Models.py:603-616)
    tgt = Pm
    if cfg.norm_pix_loss:
        mean = tgt.mean(-1, keepdim=True)
        std = (tgt.var(-1, keepdim=True) + 1.0e-6) ** 0.5
        tgt = (tgt - mean) / std
    tap("target", tgt)
    per_tok = ((pred - tgt) ** 2).mean(-1)
    loss = (per_tok * mask_t).sum() / mask_t.sum()

    # 12. recons (Models.py:618-625)
    mask_img = unpatchify(mask_t.unsqueeze(2).repeat(1, 1, cfg.patch_dim), cfg)
    p2 = pred * std + mean if cfg.norm_pix_loss else pred
    pred_img = unpatchify(p2, cfg)
    return loss, pred_img, mask_img


def encode_unmasked(P: dict, cfg: OracleConfig, imgs: torch.Tensor) -> torch.Tensor:
    """DualViT / HSIViT `forward_encoder` (Models.py:869-894, 1119-1146): every token kept, natural order.
    Same arithmetic as the masked encoder with the full (T, L) grid; increasing noise makes ids_keep the identity."""
    N = imgs.shape[0]
    taps = {}
    n1 = np.tile(np.arange(cfg.T, dtype=np.float32), (N, 1))
    n2 = np.tile(np.arange(cfg.L, dtype=np.float32), (N, 1))
    forward(P, cfg, imgs, n1, n2, cfg.T, cfg.L, taps)           # decoder / loss results are unused (nothing is masked)
    assert bool((taps["ids_keep"] == torch.arange(cfg.T * cfg.L)).all())
    return taps["latent"]


def dualvit_classify(P: dict, cfg: OracleConfig, imgs: torch.Tensor):
    """DualViT.forward(imgs) in eval mode (Models.py:975-977, head 'AGG' :962-970) -> (class_pred, pooled)."""
    lat = encode_unmasked(P, cfg, imgs)
    N = lat.shape[0]
    x = lat.reshape(N, cfg.T, cfg.L, cfg.embed_dim).permute(0, 2, 1, 3).reshape(N, cfg.L, -1).mean(1)
    return F.linear(x, P["cls_head.weight"], P["cls_head.bias"]), x


def drop_rates(cfg: OracleConfig, drop_path: float) -> list:
    """DropPath probability of every encoder block in execution order (Models.py:687-731: dpr = linspace(0, p, depth);
    blocks_1[i] and blocks_2[i] use dpr[i], blocks[j] uses dpr[s_depth + j])."""
    dpr = [x.item() for x in torch.linspace(0, drop_path, cfg.depth)]
    out = []
    if cfg.s_depth > 0:
        out += dpr[:cfg.s_depth] + dpr[:cfg.s_depth]
    if cfg.s_depth < 12:
        out += dpr[cfg.s_depth:cfg.depth]
    return out


def draw_drop_factors(cfg: OracleConfig, drop_path: float, N: int, len_t: int, len_l: int, generator=None) -> list:
    """The DropPath factors one encoder pass draws, in the reference's order (attn then mlp of every block; blocks
    with probability 0 are nn.Identity and draw nothing, Models.py:298): bernoulli_(keep) / keep on a
    [sequences, 1, 1] tensor (Models.py:245-249)."""
    nseq = ([N * len_t] * cfg.s_depth + [N * len_l] * cfg.s_depth if cfg.s_depth > 0 else []) + \
           ([N] * (cfg.depth - cfg.s_depth) if cfg.s_depth < 12 else [])
    out = []
    for p, n in zip(drop_rates(cfg, drop_path), nseq):
        if p == 0.0:
            out.append((None, None))
            continue
        keep = 1 - p
        pair = []
        for _ in range(2):
            m = torch.empty(n, 1, 1).bernoulli_(keep, generator=generator)
            if keep > 0.0:
                m.div_(keep)
            pair.append(m.reshape(-1))
        out.append(tuple(pair))
    return out


def dualvit_train_step(P: dict, cfg: OracleConfig, imgs, imgs_u, targets, lamda, noise_1, noise_2, len_t, len_l,
                       drops_cls=None, drops_rec=None):
    """One fine-tuning step of the reference (Model_Finetuning.py:150-156) up to `loss.backward()`:
    DualViT.forward(imgs, imgs_u) (Models.py:975-991) = classification branch on the unmasked encoder +
    reconstruction branch on concat(imgs, imgs_u); loss = lamda * loss_rec + CrossEntropy(ignore_index=0).
    Returns (loss_rec, class_pred, loss, grads)."""
    frozen = ("pos_embed", "decoder_pos_embed")
    Pg = {}
    for k, v in P.items():
        t = v.detach().clone()
        if k not in frozen and k != "mask_token":
            t.requires_grad_(True)
        Pg[k] = t
    N = imgs.shape[0]
    taps = {}
    n1 = np.tile(np.arange(cfg.T, dtype=np.float32), (N, 1))
    n2 = np.tile(np.arange(cfg.L, dtype=np.float32), (N, 1))
    forward(Pg, cfg, imgs, n1, n2, cfg.T, cfg.L, taps, drops_cls)
    lat = taps["latent"]
    x = lat.reshape(N, cfg.T, cfg.L, cfg.embed_dim).permute(0, 2, 1, 3).reshape(N, cfg.L, -1).mean(1)
    class_pred = F.linear(x, Pg["cls_head.weight"], Pg["cls_head.bias"])
    imgs_all = torch.cat([imgs, imgs_u], dim=0)
    loss_rec, _, _ = forward(Pg, cfg, imgs_all, noise_1, noise_2, len_t, len_l, None, drops_rec)
    loss_cls = F.cross_entropy(class_pred, targets, reduction="mean", ignore_index=0)
    loss = lamda * loss_rec + loss_cls
    loss.backward()
    grads = {k: v.grad for k, v in Pg.items() if v.grad is not None}
    return loss_rec.detach(), class_pred.detach(), loss.detach(), grads


def forward_backward(P: dict, cfg: OracleConfig, imgs, noise_1, noise_2, len_t, len_l, taps=None):
    """Forward + autograd backward of the restatement. Returns (loss, pred, mask, grads)."""
    frozen = ("pos_embed", "decoder_pos_embed")
    Pg = {}
    for k, v in P.items():
        t = v.detach().clone()
        if k not in frozen and k != "mask_token":
            t.requires_grad_(True)
        Pg[k] = t
    loss, pred, mask = forward(Pg, cfg, imgs, noise_1, noise_2, len_t, len_l, taps)
    loss.backward()
    grads = {k: v.grad for k, v in Pg.items() if v.grad is not None}
    return loss.detach(), pred.detach(), mask.detach(), grads


# --------------------------------------------------------------------------- helpers
def init_state(cfg: OracleConfig, seed: int = 0, std: float = 0.02, dtype=torch.float32) -> dict:
    """A deterministic, reference-*shaped* parameter set for GPU-box tests.

    Not the reference's init RNG order (that is pinned through fixtures); this only
    needs the right names/shapes and non-degenerate values.
    """
    g = torch.Generator().manual_seed(seed)
    D, Dd, T, L = cfg.embed_dim, cfg.decoder_embed_dim, cfg.T, cfg.L
    P = {}
    P["pos_embed"] = torch.from_numpy(sincos_pos_embed_3d(D, T, cfg.grid)).unsqueeze(0)
    P["mask_token"] = torch.zeros(1, 1, Dd)
    P["decoder_pos_embed"] = torch.from_numpy(sincos_pos_embed_3d(Dd, T, cfg.grid)).unsqueeze(0)
    P["patch_embed.proj.weight"] = torch.randn(D, 1, cfg.b_patch_size, cfg.patch_size, cfg.patch_size, generator=g) * 0.5
    P["patch_embed.proj.bias"] = torch.randn(D, generator=g) * 0.1

    def blk(pre, d):
        h = swiglu_hidden(d, cfg.mlp_ratio)
        P[f"{pre}.norm1.weight"] = 1 + 0.1 * torch.randn(d, generator=g)
        P[f"{pre}.norm1.bias"] = 0.1 * torch.randn(d, generator=g)
        for n in ("q", "k", "v", "proj"):
            P[f"{pre}.attn.{n}.weight"] = torch.randn(d, d, generator=g) * std
            P[f"{pre}.attn.{n}.bias"] = torch.randn(d, generator=g) * 0.05
        P[f"{pre}.norm2.weight"] = 1 + 0.1 * torch.randn(d, generator=g)
        P[f"{pre}.norm2.bias"] = 0.1 * torch.randn(d, generator=g)
        P[f"{pre}.mlp.w1.weight"] = torch.randn(h, d, generator=g) * std
        P[f"{pre}.mlp.w1.bias"] = torch.randn(h, generator=g) * 0.05
        P[f"{pre}.mlp.w2.weight"] = torch.randn(d, h, generator=g) * std
        P[f"{pre}.mlp.w2.bias"] = torch.randn(d, generator=g) * 0.05
        P[f"{pre}.mlp.w3.weight"] = torch.randn(h, d, generator=g) * std
        P[f"{pre}.mlp.w3.bias"] = torch.randn(h, generator=g) * 0.05

    if cfg.s_depth > 0:
        for i in range(cfg.s_depth):
            blk(f"blocks_1.{i}", D)
        for i in range(cfg.s_depth):
            blk(f"blocks_2.{i}", D)
    if cfg.s_depth < 12:
        for i in range(cfg.depth - cfg.s_depth):
            blk(f"blocks.{i}", D)
    P["norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
    P["norm.bias"] = 0.1 * torch.randn(D, generator=g)
    P["decoder_embed.weight"] = torch.randn(Dd, D, generator=g) * std
    P["decoder_embed.bias"] = torch.randn(Dd, generator=g) * 0.05
    for i in range(cfg.decoder_depth):
        blk(f"decoder_blocks.{i}", Dd)
    P["decoder_norm.weight"] = 1 + 0.1 * torch.randn(Dd, generator=g)
    P["decoder_norm.bias"] = 0.1 * torch.randn(Dd, generator=g)
    P["decoder_pred.weight"] = torch.randn(cfg.patch_dim, Dd, generator=g) * std
    P["decoder_pred.bias"] = torch.randn(cfg.patch_dim, generator=g) * 0.05
    return {k: v.to(dtype) for k, v in P.items()}


def flops_per_sample(cfg: OracleConfig, len_t: int, len_l: int) -> float:
    """Algorithmic fwd+bwd FLOPs per sample, SURVEY.md 8(a) closed form."""
    T, L, D, Dd = cfg.T, cfg.L, cfg.embed_dim, cfg.decoder_embed_dim
    TL, K = T * L, len_t * len_l
    Hm, Hd = swiglu_hidden(D, cfg.mlp_ratio), swiglu_hidden(Dd, cfg.mlp_ratio)
    nfus = cfg.depth - cfg.s_depth if cfg.s_depth < 12 else 0
    sd = cfg.s_depth
    PE = TL * cfg.patch_dim * D
    ENCl = K * (4 * D * D + 3 * D * Hm) * (2 * sd + nfus)
    ENCa = K * 2 * D * (sd * len_l + sd * len_t + nfus * K)
    DE = K * D * Dd
    DECl = TL * (4 * Dd * Dd + 3 * Dd * Hd) * cfg.decoder_depth
    DECa = TL * 2 * Dd * TL * cfg.decoder_depth
    PRED = TL * Dd * cfg.patch_dim
    fwd = 2 * (PE + ENCl + ENCa + DE + DECl + DECa + PRED)
    return fwd + 2 * fwd - 2 * PE
