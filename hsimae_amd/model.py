"""`HSIMAE` — the reference's masked-autoencoder module surface on hand-written gfx950 kernels.

Drop-in for `Models.HSIMAE` (reference Models.py:309-634) on the *pretraining* path: same constructor
signature and defaults, same submodule / parameter tree (so `state_dict()` has the reference's 535 keys,
shapes and dtypes and a checkpoint loads into the reference's `Model_Finetuning.py` unchanged), same
`forward(imgs, mask_ratio) -> (loss, pred, mask)` contract, gradients left on the same Parameter objects.

The nn.Conv3d / nn.Linear / nn.LayerNorm children are parameter containers only; all arithmetic runs in
`libhsimae_hip.so` (see include/hsimae_hip.h).  There is no CPU or eager-PyTorch fallback: tensors that are
not on a GPU, or a missing library, raise.
"""
from __future__ import annotations

import ctypes as C
import itertools
import random

import numpy as np
import torch
import torch.nn as nn

from . import _lib

__all__ = ["HSIMAE", "swiglu_hidden", "sincos_table"]


def swiglu_hidden(dim: int, mlp_ratio: float) -> int:
    """Hidden width of the gated MLP as the reference wires it (Models.py:225 via 300-301)."""
    hidden = int(dim * mlp_ratio)
    return int(mlp_ratio * ((2 * hidden // 3 + mlp_ratio - 1) // mlp_ratio))


def _sincos_axis(width: int, coords: np.ndarray) -> np.ndarray:
    freq = np.arange(width // 2, dtype=np.float32)
    freq /= width / 2.0
    freq = 1.0 / 10000 ** freq
    phase = np.einsum("m,d->md", coords.reshape(-1), freq)
    return np.concatenate([np.sin(phase), np.cos(phase)], axis=1)


def sincos_table(dim: int, t_size: int, grid: int) -> torch.Tensor:
    """Frozen 3-D sin-cos position table [1, t_size*grid^2, dim] (reference Models.py:11-47):
    first half of the channels encodes the spectral index, second half the (w, h) grid position."""
    assert dim % 4 == 0
    half = dim // 2
    gw, gh = np.meshgrid(np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32))
    spatial = np.concatenate([_sincos_axis(half // 2, gw), _sincos_axis(half // 2, gh)], axis=1)
    spectral = _sincos_axis(half, np.arange(t_size, dtype=np.float32))
    full = np.concatenate([np.repeat(spectral[:, None, :], grid * grid, axis=1),
                           np.repeat(spatial[None, :, :], t_size, axis=0)], axis=-1)
    return torch.tensor(full.reshape(-1, dim), dtype=torch.float).unsqueeze(0)


# ----------------------------------------------------------------------------- parameter containers
class PatchEmbed(nn.Module):
    """Holds `proj` = Conv3d(in_chans, D, k=s=(b_patch, p, p)) (reference Models.py:104-149)."""

    def __init__(self, img_size, patch_size, bands, b_patch_size, in_chans, embed_dim):
        super().__init__()
        assert img_size % patch_size == 0
        assert bands % b_patch_size == 0
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.bands, self.b_patch_size = bands, b_patch_size
        self.grid_size = img_size // patch_size
        self.b_grid_size = bands // b_patch_size
        self.input_size = (self.b_grid_size, self.grid_size, self.grid_size)
        self.num_patches = self.b_grid_size * self.grid_size ** 2
        print(f"img_size {self.img_size} patch_size {self.patch_size} bands {bands} b_patch_size {b_patch_size}")
        ks = [b_patch_size, patch_size, patch_size]
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=ks, stride=ks)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class SwiGLU(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.w1 = nn.Linear(dim, hidden, bias=True)
        self.w2 = nn.Linear(hidden, dim, bias=True)
        self.w3 = nn.Linear(dim, hidden, bias=True)


class Block(nn.Module):
    """Pre-LN residual block parameters: norm1, attn, norm2, mlp (reference Models.py:269-301)."""

    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = SwiGLU(dim, swiglu_hidden(dim, mlp_ratio))


class _FwdState(dict):
    """What one forward pass leaves behind for its backward: the `hsimae_io` block, the tensors it points to, and a
    lease on the workspace arena that holds the saved activations.  The arena goes back to the model's pool when the
    state is released (after the backward) or dropped; every arena carries a generation number, so a backward whose
    arena has been handed to a later forward raises instead of reading someone else's activations."""

    def release(self):
        ws, pool = self.pop("_ws", None), self.pop("_pool", None)
        if ws is not None and pool is not None:
            pool.give_back(ws)

    def check_alive(self):
        if self.get("_done"):
            raise RuntimeError("hsimae_amd: this forward pass has already been back-propagated; its saved activations are "
                               "consumed in place by the backward kernels (run the forward again instead of retain_graph)")
        ws = self.get("_ws_ref")
        if ws is None or getattr(ws, "_hs_gen", None) != self.get("_gen"):
            raise RuntimeError(
                "hsimae_amd: the activations of this forward pass are gone (its workspace was released by an earlier "
                "backward and reused by a later forward).  Call backward once per forward, or keep the graph's forward "
                "and backward adjacent when using retain_graph.")

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class _WorkspacePool:
    """Workspace arenas (hsimae_workspace_bytes each, 19.7 GB at C2) leased per forward pass.  A training step reuses one
    arena forever; `(m(x1)[0] + m(x2)[0]).backward()`, a monitoring forward between forward and backward, or deferred
    backwards simply hold several at once, as the reference's autograd graph would hold several sets of activations."""

    MAX_FREE = 2

    def __init__(self):
        self.free, self.gen = [], 0

    def lease(self, nbytes, device):
        best = None
        for t in self.free:
            if t.device == device and t.numel() >= nbytes and (best is None or t.numel() < best.numel()):
                best = t
        if best is not None:
            self.free = [t for t in self.free if t is not best]
        else:
            if self.free:                                  # too small / other device: let the allocator have it back first
                self.free.pop(0)
            best = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.gen += 1
        best._hs_gen = self.gen
        return best

    def give_back(self, t):
        self.free.append(t)
        while len(self.free) > self.MAX_FREE:
            self.free.pop(0)


class _Step(torch.autograd.Function):
    """Whole forward / whole backward as one autograd node; parameter grads are written by the kernels
    into the model's flat gradient buffer and attached to the Parameters directly."""

    @staticmethod
    def forward(ctx, anchor, model, imgs, ratio, noise, grid):
        loss, pred, mask, state = model._run_forward(imgs, ratio, noise, grid, want_latent=False)
        ctx.model, ctx.state = model, state
        ctx.mark_non_differentiable(pred, mask)
        return loss, pred, mask

    @staticmethod
    def backward(ctx, g_loss, g_pred, g_mask):
        ctx.model._run_backward(ctx.state, g_loss)
        return None, None, None, None, None, None


class _EncodeFn(torch.autograd.Function):
    """forward_encoder (Models.py:537-571) as an autograd node: hsimae_encode / hsimae_encode_backward."""

    @staticmethod
    def forward(ctx, anchor, model, x, ratio, noise, grid):
        _, _, _, st = model._run_forward(x, ratio, noise, grid, want_latent=True, encoder_only=True)
        ctx.model, ctx.state = model, st
        ids_r, ids_k = st["ids_restore"].long(), st["ids_keep"].long()
        ctx.mark_non_differentiable(st["mask"], ids_r, ids_k)
        return st["latent"], st["mask"], ids_r, ids_k

    @staticmethod
    def backward(ctx, g_lat, *_):
        ctx.model._encode_backward(ctx.state, g_lat)
        return None, None, None, None, None, None


class _DecodeFn(torch.autograd.Function):
    """forward_decoder (Models.py:573-601) as an autograd node: hsimae_decode / hsimae_decode_backward."""

    @staticmethod
    def forward(ctx, anchor, latent, model, ids_restore):
        pred, st = model._run_decode(latent, ids_restore)
        ctx.model, ctx.state = model, st
        return pred

    @staticmethod
    def backward(ctx, g_pred):
        return None, ctx.model._decode_backward(ctx.state, g_pred), None, None


class _LossFn(torch.autograd.Function):
    """forward_loss (Models.py:603-616) as an autograd node: the loss kernel also emits dLoss/dpred."""

    @staticmethod
    def forward(ctx, pred, model, imgs, mask):
        loss, dpred = model._run_loss(imgs, pred, mask, want_grad=True)
        ctx.save_for_backward(dpred)
        ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return (dpred[:, :72].float() * g).view(ctx.shape), None, None, None


class HSIMAE(nn.Module):
    """Masked autoencoder for 9x9xB hyperspectral cubes, MI355X-native."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.0,
                 norm_layer=nn.LayerNorm, norm_pix_loss=False, bands=16, b_patch_size=4, no_qkv_bias=False,
                 trunc_init=False, s_depth=6, **kwargs):
        super().__init__()
        self.dim, self.dec_dim = embed_dim, decoder_embed_dim
        self.s_depth, self.depth = s_depth, depth
        self.b_pred_patch_size = b_patch_size
        self.trunc_init, self.norm_pix_loss = trunc_init, norm_pix_loss
        self.num_heads, self.decoder_num_heads = num_heads, decoder_num_heads
        self.mlp_ratio, self.in_chans = mlp_ratio, in_chans
        self._norm_layer, self._no_qkv_bias = norm_layer, no_qkv_bias

        self.patch_embed = PatchEmbed(img_size, patch_size, bands, b_patch_size, in_chans, embed_dim)
        self.input_size = self.patch_embed.input_size
        n_tok = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, n_tok, embed_dim))

        def stack(n, dim, heads):
            return nn.ModuleList([Block(dim, heads, mlp_ratio, not no_qkv_bias, norm_layer) for _ in range(n)])

        if s_depth > 0:
            self.blocks_1 = stack(s_depth, embed_dim, num_heads)
            self.blocks_2 = stack(s_depth, embed_dim, num_heads)
        if s_depth < 12:                                   # the reference hard-codes 12 here (Models.py:385)
            self.blocks = stack(max(0, depth - s_depth), embed_dim, num_heads)
        self.norm = norm_layer(embed_dim)
        if kwargs.get("_num_class") is not None:           # DualViT registers its head here (Models.py:742)
            self.cls_head = nn.Linear(embed_dim * self.patch_embed.b_grid_size, kwargs["_num_class"])
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))   # kept for the wire format; unused
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, n_tok, decoder_embed_dim))
        self.decoder_blocks = stack(decoder_depth, decoder_embed_dim, decoder_num_heads)
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, b_patch_size * patch_size ** 2 * in_chans, bias=True)
        self.initialize_weights()

        # runtime state (not part of the state_dict)
        self._flat = self._flat_grad = self._wpk = self._pack_table = None
        self._pool = _WorkspacePool()
        self._cfg = None
        self._precision = _lib.PREC_BF16
        self._deterministic = None
        self._det_acc = None
        self._packed_version = -1
        self._anchor = None
        self._reducer = None
        self._world_override = None     # see _dp_world
        self._grads_home = [False, False]
        self.want_recons = True
        self.len_t = self.len_l = None
        self._last_imgs = None
        print("model initialized")

    # ------------------------------------------------------------------ init (reference Models.py:429-459)
    def initialize_weights(self):
        t, g = self.input_size[0], self.input_size[1]
        self.pos_embed.data.copy_(sincos_table(self.dim, t, g))
        self.decoder_pos_embed.data.copy_(sincos_table(self.dec_dim, t, g))
        self.pos_embed.requires_grad = False
        self.decoder_pos_embed.requires_grad = False
        w = self.patch_embed.proj.weight.data
        if self.trunc_init:
            nn.init.trunc_normal_(w)
            nn.init.trunc_normal_(self.mask_token, std=0.02)
        else:
            nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
            nn.init.normal_(self.mask_token, std=0.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            if self.trunc_init:
                nn.init.trunc_normal_(m.weight, std=0.02)
            else:
                nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------ pure index maps (Models.py:461-482)
    def patchify(self, imgs):
        N, _, T, H, W = imgs.shape
        p, u = self.patch_embed.patch_size[0], self.b_pred_patch_size
        assert H == W and H % p == 0 and T % u == 0
        h = w = H // p
        t = T // u
        x = imgs.reshape(N, t, u, h, p, w, p).permute(0, 1, 3, 5, 2, 4, 6)
        self.patch_info = (N, T, H, W, p, u, t, h, w)
        return x.reshape(N, t * h * w, u * p * p)

    def unpatchify(self, x):
        N, T, H, W, p, u, t, h, w = self.patch_info
        x = x.reshape(N, t, h, w, u, p, p).permute(0, 1, 4, 2, 5, 3, 6)
        return x.reshape(N, 1, T, H, W)

    # ------------------------------------------------------------------ host-side grid choice (Models.py:484-493)
    _cand_cache: dict = {}

    @classmethod
    def grid_candidates(cls, T, L, mask_ratio):
        key = (T, L, float(mask_ratio))
        if key not in cls._cand_cache:
            pairs = list(itertools.product(range(2, T + 1), range(2, L + 1)))
            target = (1 - mask_ratio) * T * L
            dist = abs(target - torch.tensor([a * b for a, b in pairs]))      # fp32, as the reference computes it
            best = torch.where(dist == torch.min(dist))[0].tolist()
            cls._cand_cache[key] = [pairs[i] for i in best]
        return cls._cand_cache[key]

    def get_dim_patches(self, T, L, mask_ratio):
        """One python-`random` draw per call, even with a single candidate (keeps the RNG stream)."""
        cands = self.grid_candidates(T, L, mask_ratio)
        return cands[random.sample(range(len(cands)), 1)[0]]

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        if set_to_none:
            self._grads_home = [False, False]

    # ------------------------------------------------------------------ device-side plumbing
    def _check_supported(self):
        pe = self.patch_embed
        if (pe.img_size[0] != 9 or pe.patch_size[0] != 3 or self.b_pred_patch_size != 8 or self.in_chans != 1):
            raise NotImplementedError("hsimae_amd kernels are built for img_size=9, patch_size=3, b_patch_size=8, in_chans=1")
        if self._no_qkv_bias:
            raise NotImplementedError("no_qkv_bias=True is not supported by the gfx950 kernels")
        if self._norm_layer is not nn.LayerNorm:
            raise NotImplementedError("only norm_layer=nn.LayerNorm is supported by the gfx950 kernels")

    def _config(self) -> _lib.Config:
        if self._cfg is None:
            self._check_supported()
            depth = self.depth
            self._cfg = _lib.Config(
                bands=self.patch_embed.bands, embed_dim=self.dim, depth=depth, s_depth=self.s_depth,
                num_heads=self.num_heads, dec_dim=self.dec_dim, dec_depth=len(self.decoder_blocks),
                dec_heads=self.decoder_num_heads, hidden=swiglu_hidden(self.dim, self.mlp_ratio),
                dec_hidden=swiglu_hidden(self.dec_dim, self.mlp_ratio), norm_pix_loss=int(bool(self.norm_pix_loss)),
                precision=self._precision)
        return self._cfg

    def set_precision(self, precision="bf16"):
        """GEMM operand type of the encoder blocks' linears: "bf16" (default) or "fp8" (MX block-scaled e4m3 MFMA,
        include/hsimae_hip.h `hsimae_config.precision`).  Compute copies only: parameters, gradients and the state_dict
        stay fp32.  fp8 is applied where it pays — embed_dim >= 512 (K = 128 per MX MFMA at twice the bf16 rate needs deep
        contractions to beat the quantisation of the activations; measured: Huge 49 vs 60 ms per step, Large 37.8 vs 37.2) —
        and below that width the bf16 kernels run, so "fp8" is never slower than "bf16" and at embed_dim 128 / 256 computes
        exactly what "bf16" computes.  HSIMAE_FP8_UNFUSED=1 forces every encoder linear onto the MX GEMMs, layer at a time,
        at any width (tests of the generic path)."""
        prec = {"bf16": _lib.PREC_BF16, "fp8": _lib.PREC_FP8}[precision]
        if prec != self._precision:
            self._precision, self._cfg = prec, None
            self._wpk = self._pack_table = None          # the packed images are laid out per precision
            self._packed_version = -1
        return self

    def _plist(self):
        return [p for n, p in self.named_parameters() if not n.startswith("cls_head.")]

    def _ensure_flat(self, device):
        """Re-home every Parameter as a view of one flat fp32 buffer in registration order (the layout
        hsimae_param_layout() defines) so kernels address parameters and gradients by offset."""
        params = self._plist()
        if (self._flat is not None and self._flat.device == device and
                params[0].data_ptr() == self._flat.data_ptr() and
                params[-1].data_ptr() == self._flat.data_ptr() + 4 * (self._flat.numel() - params[-1].numel())):
            if self._wpk is None:
                self._ensure_pack_buffers(device)
            return
        lib, cfg = _lib.load(), self._config()
        n = len(params)
        offs, sizes = (C.c_int64 * n)(), (C.c_int64 * n)()
        cnt = lib.hsimae_param_layout(C.byref(cfg), offs, sizes, n)
        if cnt < 0:
            _lib.check(cnt, "hsimae_param_layout")
        if cnt != n or any(sizes[i] != params[i].numel() for i in range(n)):
            raise RuntimeError("parameter tree does not match the kernel library's layout")
        total = offs[n - 1] + sizes[n - 1]
        flat = torch.empty(total, dtype=torch.float32, device=device)
        for i, p in enumerate(params):
            if p.dtype != torch.float32:
                raise RuntimeError("hsimae_amd keeps fp32 master parameters; found " + str(p.dtype))
            view = flat[offs[i]: offs[i] + sizes[i]].view(p.shape)
            view.copy_(p.data)
            p.data = view
        self._flat, self._offs, self._sizes = flat, list(offs), list(sizes)
        self._flat_grad = torch.zeros_like(flat)
        self._grads_home = [False, False]
        self._flat_scratch = torch.zeros_like(flat)
        self._grad_views = [self._flat_grad[o: o + s].view(p.shape) for o, s, p in zip(self._offs, self._sizes, params)]
        self._trainable = [i for i, p in enumerate(params) if p.requires_grad and i != 1]   # 1 = mask_token (never used)
        # the encoder's / decoder's parameters are two contiguous ranges of the flat buffer (stand-alone backward passes)
        dec0 = next(i for i, p in enumerate(params) if p is self.norm.bias) + 1       # decoder_embed.weight follows norm.bias
        self._enc_idx = [i for i in self._trainable if i < dec0]
        self._dec_idx = [i for i in self._trainable if i >= dec0]
        self._enc_range = (self._offs[self._enc_idx[0]], self._offs[dec0])
        self._dec_range = (self._offs[dec0], total)
        self._params_cache = params
        self._anchor = torch.zeros((), device=device, requires_grad=True)
        self._ensure_pack_buffers(device)

    def _ensure_pack_buffers(self, device):
        lib, cfg = _lib.load(), self._config()
        wpk_elems = lib.hsimae_wpk_elems(C.byref(cfg))
        if wpk_elems < 0:
            raise NotImplementedError("hsimae_amd: this configuration is not supported by the gfx950 kernels "
                                      "(widths must be multiples of 8 up to 512, head dim 8 or 16)")
        self._wpk = torch.zeros(wpk_elems, dtype=torch.bfloat16, device=device)
        tb = lib.hsimae_pack_table_bytes(C.byref(cfg))
        host = torch.empty(tb, dtype=torch.uint8)
        _lib.check(lib.hsimae_build_pack_table(C.byref(cfg), self._flat.data_ptr(), self._wpk.data_ptr(), host.data_ptr()),
                   "hsimae_build_pack_table")
        self._pack_table = host.to(device)
        self._packed_version = -1

    def _ensure_packed(self, stream):
        ver = sum(p._version for p in self._params_cache)
        if ver != self._packed_version:
            _lib.check(_lib.load().hsimae_pack_params(C.byref(self._config()), self._pack_table.data_ptr(), stream),
                       "hsimae_pack_params")
            self._packed_version = ver

    # ------------------------------------------------------------------ forward / backward drivers
    def _run_forward(self, imgs, mask_ratio, noise, grid, want_latent, encoder_only=False, drop_scale=None, ws_slot=0):
        """One hsimae_forward / hsimae_encode call.  Returns (loss, pred_img, mask_img, state); `state` holds the lease on
        the workspace arena with the saved activations until `state.release()` or until it is dropped."""
        if not imgs.is_cuda:
            raise RuntimeError("hsimae_amd.HSIMAE runs on MI355X only (no CPU fallback): move the model and inputs to a GPU")
        if imgs.dim() != 5 or imgs.shape[1] != 1 or imgs.shape[2] != self.patch_embed.bands or imgs.shape[3:] != (9, 9):
            raise ValueError(f"expected imgs [N,1,{self.patch_embed.bands},9,9], got {tuple(imgs.shape)}")
        if imgs.dtype != torch.float32:
            imgs = imgs.float()
        dev = imgs.device
        with torch.cuda.device(dev):                  # the library's side stream / events belong to the current device
            return self._run_forward_on(dev, imgs, mask_ratio, noise, grid, want_latent, encoder_only, drop_scale)

    def _run_forward_on(self, dev, imgs, mask_ratio, noise, grid, want_latent, encoder_only, drop_scale):
        lib, cfg = _lib.load(), self._config()
        self._ensure_flat(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        self._ensure_packed(stream)
        N, T, L = imgs.shape[0], self.input_size[0], self.input_size[1] ** 2
        # RNG order of the reference: python random (grid), then rand(N,T), then rand(N,L)  (Models.py:501-513)
        len_t, len_l = grid if grid is not None else self.get_dim_patches(T, L, mask_ratio)
        if noise is None:
            n1 = torch.rand(N, T, device=dev)
            n2 = torch.rand(N, L, device=dev)
        else:
            n1 = noise[0].to(device=dev, dtype=torch.float32).contiguous()
            n2 = noise[1].to(device=dev, dtype=torch.float32).contiguous()
        self.len_t, self.len_l = int(len_t), int(len_l)
        K, TL = self.len_t * self.len_l, T * L
        self.patch_embed.output_size = torch.Size([N, T, L, self.dim])
        self.patch_info = (N, imgs.shape[2], 9, 9, 3, 8, T, 3, 3)
        self._last_imgs = imgs

        nbytes = lib.hsimae_workspace_bytes(C.byref(cfg), N, self.len_t, self.len_l)
        if nbytes < 0:
            raise RuntimeError("hsimae_workspace_bytes: unsupported configuration")
        ws = self._lease_workspace(nbytes, dev, N, K)
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        loss = torch.empty((), dtype=torch.float32, device=dev)
        mask = torch.empty(N, TL, dtype=torch.float32, device=dev)
        ids_keep = torch.empty(N, K, dtype=torch.int32, device=dev)
        ids_restore = torch.empty(N, TL, dtype=torch.int32, device=dev)
        pred_img = mask_img = None
        if self.want_recons and not encoder_only:
            pred_img = torch.empty(N, 1, imgs.shape[2], 9, 9, dtype=torch.float32, device=dev)
            mask_img = torch.empty_like(pred_img)
        latent = torch.empty(N, K, self.dim, dtype=torch.float32, device=dev) if want_latent else None
        world = self._dp_world()
        io = _lib.IO(
            x=imgs.data_ptr(), sn=imgs.stride(0), sb=imgs.stride(2), sh=imgs.stride(3), sw=imgs.stride(4),
            N=N, len_t=self.len_t, len_l=self.len_l, noise1=n1.data_ptr(), noise2=n2.data_ptr(),
            params=self._flat.data_ptr(), wpk=self._wpk.data_ptr(), workspace=ws_ptr, workspace_bytes=nbytes,
            grad_scale=1.0 / world, want_recons=int(pred_img is not None), loss=loss.data_ptr(),
            pred_img=_lib.ptr(pred_img), mask_img=_lib.ptr(mask_img), mask=mask.data_ptr(),
            ids_keep=ids_keep.data_ptr(), ids_restore=ids_restore.data_ptr(), latent=_lib.ptr(latent), pred=None,
            drop_scale=_lib.ptr(drop_scale), bucket_stream=None)
        state = _FwdState(io=io, keep=(imgs, n1, n2, mask, ids_keep, ids_restore, drop_scale, self._flat, self._wpk),
                          latent=latent, ids_keep=ids_keep, ids_restore=ids_restore, mask=mask,
                          _ws=ws, _ws_ref=ws, _gen=ws._hs_gen, _pool=self._pool, grid=(self.len_t, self.len_l))
        if encoder_only:
            _lib.check(lib.hsimae_encode(C.byref(cfg), C.byref(io), stream), "hsimae_encode")
        else:
            _lib.check(lib.hsimae_forward(C.byref(cfg), C.byref(io), stream), "hsimae_forward")
        if pred_img is None:
            pred_img = torch.empty(0, device=dev)
            mask_img = torch.empty(0, device=dev)
        return loss, pred_img, mask_img, state

    def _lease_workspace(self, nbytes, dev, N, K):
        """Arena for one pass.  Widths that are not multiples of 32 are stored zero-padded (include/hsimae_hip.h,
        hsimae_io.workspace): the arena is zero-filled whenever its carve (a function of N and K) changes; the kernels
        keep the pad columns zero from then on."""
        ws = self._pool.lease(nbytes + 256, dev)
        if self.dim % 32 or self.dec_dim % 32:
            key = (N, K, self._precision)
            if getattr(ws, "_hs_carve", None) != key:
                ws.zero_()
                ws._hs_carve = key
        return ws

    def _apply_grads(self, scratch, scale, idxs=None, rng=None):
        """`.grad` semantics for the parameters `idxs` (flat range `rng`): scratch * scale is ASSIGNED where the gradient
        was cleared (None) and ACCUMULATED where one exists — per parameter, whatever tensor `.grad` currently is."""
        params, views = self._params_cache, self._grad_views
        idxs = self._trainable if idxs is None else idxs
        lo, hi = (0, scratch.numel()) if rng is None else rng
        cur = [params[i].grad for i in idxs]
        # which of the two parameter ranges (encoder, decoder) have every `.grad` attached to its flat view after this call:
        # FusedAdamW's O(1) check (optim.py _sync_grads); HSIMAE.zero_grad clears it
        full = idxs is self._trainable
        enc, dec = full or idxs is self._enc_idx, full or idxs is self._dec_idx
        if all(g is None for g in cur):
            torch.mul(scratch[lo:hi], scale, out=self._flat_grad[lo:hi])
            for i in idxs:
                params[i].grad = views[i]
            self._grads_home = [self._grads_home[0] or enc, self._grads_home[1] or dec]
        elif all(g is views[i] or (g is not None and g.data_ptr() == views[i].data_ptr() and g.shape == views[i].shape)
                 for g, i in zip(cur, idxs)):
            self._flat_grad[lo:hi].addcmul_(scratch[lo:hi], torch.as_tensor(scale, device=scratch.device, dtype=torch.float32))
        else:                                             # mixed: some cleared, some replaced by the caller
            if enc:
                self._grads_home[0] = False
            if dec:
                self._grads_home[1] = False
            for g, i in zip(cur, idxs):
                o, n = self._offs[i], self._sizes[i]
                upd = (scratch[o:o + n] * scale).view(params[i].shape)
                if g is None:
                    views[i].copy_(upd)
                    params[i].grad = views[i]
                else:
                    g.add_(upd)

    def _call_backward(self, fn, what, state, *args):
        """Shared driver of the three backward entry points: zeroed scratch, bucket callback, reducer hand-shake."""
        state.check_alive()
        lib, cfg = _lib.load(), self._config()
        dev = self._flat.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            scratch = self._flat_scratch
            scratch.zero_()                       # weight grads are accumulated with atomics
            red = self._reducer
            io = state["io"]
            io.det_acc = self._det_buffer(dev)
            if red is not None:
                io.bucket_stream = red.launch_stream_handle(dev)
                cb = red.make_callback(scratch)
            else:
                cb = _lib.BUCKET_CB(0)
                io.bucket_stream = None
            state["_done"] = True
            try:
                _lib.check(fn(C.byref(cfg), C.byref(io), *args, scratch.data_ptr(), cb, None, stream), what)
            finally:
                if red is not None:
                    red.finish()
        return scratch

    @property
    def deterministic(self):
        """Bit-reproducible gradients: set `model.deterministic = True`, or HSIMAE_DETERMINISTIC=1, or
        torch.use_deterministic_algorithms(True).  Weight-gradient partial sums are then accumulated in 64-bit fixed point
        with integer atomics (order-independent) instead of fp32 atomics; ~5 % slower, +8 B per parameter."""
        if self._deterministic is not None:
            return self._deterministic
        import os
        return os.environ.get("HSIMAE_DETERMINISTIC", "0") not in ("", "0") or torch.are_deterministic_algorithms_enabled()

    @deterministic.setter
    def deterministic(self, on):
        self._deterministic = None if on is None else bool(on)

    def _det_buffer(self, dev):
        if not self.deterministic:
            return None
        if self._det_acc is None or self._det_acc.device != dev or self._det_acc.numel() != self._flat.numel():
            self._det_acc = torch.empty(self._flat.numel(), dtype=torch.int64, device=dev)
        return self._det_acc.data_ptr()

    def _run_backward(self, state, g_loss):
        lib = _lib.load()
        scratch = self._call_backward(lib.hsimae_backward, "hsimae_backward", state)
        # chain rule with d/d(loss) handed in by autograd (1.0 for loss.backward())
        self._apply_grads(scratch, g_loss)
        state.release()

    def _encode_backward(self, state, g_latent):
        lib = _lib.load()
        dlat = g_latent.to(torch.float32).contiguous()
        scratch = self._call_backward(lib.hsimae_encode_backward, "hsimae_encode_backward", state, dlat.data_ptr())
        self._apply_grads(scratch, 1.0, self._enc_idx, self._enc_range)
        state.release()

    def _run_decode(self, x, ids_restore):
        if not x.is_cuda:
            raise RuntimeError("hsimae_amd.HSIMAE runs on MI355X only (no CPU fallback)")
        N, K, D = x.shape
        T, L = self.input_size[0], self.input_size[1] ** 2
        if self.len_t is None or self.len_t * self.len_l != K or D != self.dim or ids_restore.shape != (N, T * L):
            raise ValueError("forward_decoder: latent does not match the grid of the last encoder call")
        dev = x.device
        lib, cfg = _lib.load(), self._config()
        with torch.cuda.device(dev):
            self._ensure_flat(dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            self._ensure_packed(stream)
            nbytes = lib.hsimae_workspace_bytes(C.byref(cfg), N, self.len_t, self.len_l)
            ws = self._lease_workspace(nbytes, dev, N, K)
            lat = x.detach().to(torch.float32).contiguous()
            ids = ids_restore.to(device=dev, dtype=torch.int32).contiguous()
            pred = torch.empty(N, T * L, 72, dtype=torch.float32, device=dev)
            world = self._dp_world()
            io = _lib.IO(N=N, len_t=self.len_t, len_l=self.len_l, params=self._flat.data_ptr(), wpk=self._wpk.data_ptr(),
                         workspace=(ws.data_ptr() + 255) // 256 * 256, workspace_bytes=nbytes, grad_scale=1.0 / world,
                         ids_restore=ids.data_ptr(), bucket_stream=None)
            state = _FwdState(io=io, keep=(lat, ids, self._flat, self._wpk), _ws=ws, _ws_ref=ws, _gen=ws._hs_gen,
                              _pool=self._pool, shape=(N, K, D))
            _lib.check(lib.hsimae_decode(C.byref(cfg), C.byref(io), lat.data_ptr(), pred.data_ptr(), stream), "hsimae_decode")
        return pred, state

    def _decode_backward(self, state, g_pred):
        lib = _lib.load()
        N, K, D = state["shape"]
        dpred = g_pred.to(torch.float32).contiguous()
        dlat = torch.empty(N, K, D, dtype=torch.float32, device=dpred.device)
        scratch = self._call_backward(lib.hsimae_decode_backward, "hsimae_decode_backward", state, dpred.data_ptr(),
                                      dlat.data_ptr())
        self._apply_grads(scratch, 1.0, self._dec_idx, self._dec_range)
        state.release()
        return dlat

    def _run_loss(self, imgs, pred, mask, want_grad=False):
        if not imgs.is_cuda:
            raise RuntimeError("hsimae_amd.HSIMAE runs on MI355X only (no CPU fallback)")
        dev = imgs.device
        lib = _lib.load()
        imgs = imgs.float()
        N, T = imgs.shape[0], self.input_size[0]
        TL = T * self.input_size[1] ** 2
        pr = pred.detach().to(torch.float32).reshape(N * TL, 72).contiguous()
        mk = mask.detach().to(torch.float32).reshape(N, TL).contiguous()
        sum_mask = float(mk.sum().item())
        with torch.cuda.device(dev):
            partial = torch.empty(lib.hsimae_loss_partials(N, T), dtype=torch.float32, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            dpred = torch.empty(N * TL, 96, dtype=torch.bfloat16, device=dev) if want_grad else None
            # data parallel: the reducer SUMs the ranks' gradients, so dLoss/dpred carries the 1/world of the mean
            # (hsimae_forward folds the same factor in through hsimae_io.grad_scale)
            world = self._dp_world()
            p = _lib.LossParams(x=imgs.data_ptr(), sn=imgs.stride(0), sb=imgs.stride(2), sh=imgs.stride(3), sw=imgs.stride(4),
                                N=N, T=T, pred=pr.data_ptr(), mask=mk.data_ptr(), norm_pix=int(bool(self.norm_pix_loss)),
                                inv_scale=(1.0 / (72.0 * sum_mask * world)) if want_grad else 0.0, partial=partial.data_ptr(),
                                loss=loss.data_ptr(), sum_mask=sum_mask, dpred=_lib.ptr(dpred))
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.hsimae_loss(C.byref(p), stream), "hsimae_loss")
        self._last_imgs = imgs
        return loss, dpred

    def _wants_grad(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    # ------------------------------------------------------------------ public API (reference Models.py:537-634)
    def forward(self, imgs, mask_ratio=0.75, noise=None, grid=None):
        """-> (loss, pred [N,1,B,9,9], mask [N,1,B,9,9]).  `noise=(noise_1 [N,T], noise_2 [N,9])` and
        `grid=(len_t, len_l)` optionally replace the RNG draws (parity tests)."""
        if self._wants_grad():
            if imgs.is_cuda:
                self._ensure_flat(imgs.device)
            anchor = self._anchor if self._anchor is not None else torch.zeros((), requires_grad=True)
            return _Step.apply(anchor, self, imgs, mask_ratio, noise, grid)
        loss, pred, mask, st = self._run_forward(imgs, mask_ratio, noise, grid, want_latent=False)
        st.release()
        return loss, pred, mask

    def forward_encoder(self, x, mask_ratio, noise=None, grid=None):
        """-> (latent [N,K,D], mask [N,TL], ids_restore [N,TL] int64, ids_keep [N,K] int64)  (Models.py:537-571).
        Differentiable w.r.t. the encoder's parameters (hsimae_encode_backward)."""
        if self._wants_grad():
            if x.is_cuda:
                self._ensure_flat(x.device)
            anchor = self._anchor if self._anchor is not None else torch.zeros((), requires_grad=True)
            return _EncodeFn.apply(anchor, self, x, mask_ratio, noise, grid)
        _, _, _, st = self._run_forward(x, mask_ratio, noise, grid, want_latent=True, encoder_only=True)
        st.release()
        return st["latent"], st["mask"], st["ids_restore"].long(), st["ids_keep"].long()

    def forward_decoder(self, x, ids_restore):
        """latent [N,K,D] + ids_restore [N,TL] -> pred [N,TL,72]  (Models.py:573-601).  K must be the len_t * len_l of the
        last `forward_encoder` / `forward` call (the kept tokens form that grid).  Differentiable w.r.t. the latent and
        the decoder's parameters (hsimae_decode_backward)."""
        if self._wants_grad() or (torch.is_grad_enabled() and x.requires_grad):
            if x.is_cuda:
                self._ensure_flat(x.device)
            anchor = self._anchor if self._anchor is not None else torch.zeros((), requires_grad=True)
            return _DecodeFn.apply(anchor, x, self, ids_restore)
        pred, st = self._run_decode(x, ids_restore)
        st.release()
        return pred

    def forward_loss(self, imgs, pred, mask):
        """Masked reconstruction loss of pred [N,TL,72] against the cube (Models.py:603-616); differentiable w.r.t. pred.
        Like the reference it leaves `mean` / `var` (per-token target statistics) for `recons`."""
        if torch.is_grad_enabled() and pred.requires_grad:
            return _LossFn.apply(pred, self, imgs, mask)
        return self._run_loss(imgs, pred, mask)[0]

    @property
    def mean(self):
        """Per-token target mean [N,TL,1] of the last cube batch (the reference stashes it in forward_loss, :609)."""
        return self.patchify(self._last_imgs).mean(dim=-1, keepdim=True)

    @property
    def var(self):
        """sqrt(var + 1e-6) [N,TL,1] of the last cube batch — the reference's `self.var` is the std (:610)."""
        return (self.patchify(self._last_imgs).var(dim=-1, keepdim=True) + 1.0e-6) ** 0.5

    def recons(self, mask, pred):
        """Models.py:618-625: token mask and (de-normalised) prediction as [N,1,B,9,9] images."""
        mask = self.unpatchify(mask.unsqueeze(2).repeat(1, 1, pred.shape[2]))
        if self.norm_pix_loss:
            pred = pred * self.var + self.mean
        return mask, self.unpatchify(pred)

    # ------------------------------------------------------------------ data parallel (not in the reference)
    def _dp_world(self) -> int:
        """The 1/world folded into dLoss/dpred: the reducer's world size; with the reducer detached, `_world_override` (bench.py's
        self-check runs the SAME arithmetic — same bf16 roundings — without the bucketed collectives) or 1."""
        if self._reducer is not None:
            return self._reducer.world_size
        return self._world_override or 1

    def enable_data_parallel(self, process_group=None, bucket_bytes=4 << 20, broadcast=True, force_collectives=False):
        """One process per GPU: average gradients over the group with bucketed RCCL all-reduce launched from
        inside the backward schedule (overlapped with the remaining backward kernels)."""
        from .parallel import GradReducer
        self._reducer = GradReducer(process_group, bucket_bytes, force_collectives)
        dev = next(self.parameters()).device
        if dev.type == "cuda" and self._reducer.world_size > 1:
            # the ranges arrive in the order the backward schedule enqueues them, and that order depends on whether the
            # library forks the two axis stacks onto a side stream (rank-local: env switch, hipStreamCreate): ranks that
            # disagree would issue different all-reduce slices.  Checked once, here.
            import torch.distributed as dist
            # (a tensor collective on the model's own device, under its device guard: object collectives stage through
            #  torch.cuda.current_device(), which is still cuda:0 on every rank of a group made without device_id — RCCL then
            #  sees one GPU twice)
            with torch.cuda.device(dev):
                mine = torch.tensor([int(_lib.load().hsimae_two_streams_active())], dtype=torch.int32, device=dev)
                gathered = [torch.zeros_like(mine) for _ in range(self._reducer.world_size)]
                dist.all_gather(gathered, mine, group=process_group)
                flags = [int(t.item()) for t in gathered]
            if len(set(flags)) != 1:
                raise RuntimeError(f"hsimae_amd: ranks disagree on the two-stream backward schedule ({flags}); set "
                                   "HSIMAE_TWO_STREAMS identically on every rank")
        if broadcast:
            dev = next(self.parameters()).device
            if dev.type == "cuda":
                self._ensure_flat(dev)
                self._reducer.broadcast(self._flat)
                self._packed_version = -1
            else:
                for p in self.parameters():
                    self._reducer.broadcast(p.data)
        return self
