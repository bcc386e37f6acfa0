"""Row N3: the dual-branch fine-tuning model (`Models.DualViT`, Models.py:637-993) on the MI355X-native
kernels — inference AND the fine-tuning step (Model_Finetuning.py:150-156).

  class_pred = cls_head(AGG(norm(encoder(imgs))))                      (forward_encoder :869-894, head :962-970)
  with imgs_u: the masked-autoencoder path on concat(imgs, imgs_u)      (forward :975-993) — the pretraining hot path
  training mode: DropPath (stochastic depth, Models.py:235-263) on both residual branches of every encoder block,
  rate linspace(0, drop_path, depth)[i] (:687), one Bernoulli factor per sequence of the block's input.

Same constructor keywords, same parameter names / order / init stream as the reference (the head is registered
between `norm` and `decoder_embed`), so checkpoints load both ways.  Everything on the encoder / decoder runs through
the HIP library: `hsimae_encode` (+ `hsimae_encode_backward`) = the masked schedule with the full token grid and
identity order, `hsimae_forward` / `hsimae_backward` for the reconstruction branch, DropPath as per-row factor vectors
(`hsimae_io.drop_scale`), `hsimae_agg_pool` + `hsimae_gemm` for the head's forward.  The head's backward (an
[N, T*D] x [classes, T*D] product) is three torch matmuls on the device.
RNG: DropPath factors are drawn from torch's generator of the input's device in the reference's order (classification
pass, then grid / masking noise, then reconstruction pass); `drop_factors=` replaces the draws (parity tests).
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .model import HSIMAE


class _FineTuneStep(torch.autograd.Function):
    """DualViT.forward in training as one autograd node: encoder / decoder parameter gradients are written by the kernels
    into the model's flat gradient buffer; the head's two parameters are ordinary autograd inputs."""

    @staticmethod
    def forward(ctx, anchor, head_w, head_b, model, imgs, imgs_u, ratio, noise, grid, drop_factors):
        out = model._train_forward(imgs, imgs_u, ratio, noise, grid, drop_factors)
        ctx.model, ctx.saved = model, out["saved"]
        if imgs_u is None:
            return out["class_pred"]
        ctx.mark_non_differentiable(out["pred_rec"], out["mask"])
        return out["loss_rec"], out["pred_rec"], out["mask"], out["class_pred"]

    @staticmethod
    def backward(ctx, *gs):
        g_loss, g_cls = (None, gs[0]) if len(gs) == 1 else (gs[0], gs[3])
        gw, gb = ctx.model._train_backward(ctx.saved, g_loss, g_cls)
        return None, gw, gb, None, None, None, None, None, None, None


class DualViT(HSIMAE):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, s_depth=6, num_heads=16,
                 mlp_ratio=4.0, norm_layer=nn.LayerNorm, bands=32, b_patch_size=8, num_class=100, no_qkv_bias=False,
                 trunc_init=False, drop_path=0., decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16,
                 norm_pix_loss=False, **kwargs):
        super().__init__(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, depth=depth,
                         num_heads=num_heads, decoder_embed_dim=decoder_embed_dim, decoder_depth=decoder_depth,
                         decoder_num_heads=decoder_num_heads, mlp_ratio=mlp_ratio, norm_layer=norm_layer,
                         norm_pix_loss=norm_pix_loss, bands=bands, b_patch_size=b_patch_size, no_qkv_bias=no_qkv_bias,
                         trunc_init=trunc_init, s_depth=s_depth, _num_class=num_class)
        self.drop_path = drop_path            # stochastic depth: active in training mode only (Models.py:244)
        self.num_class = num_class
        self._head_pack = None

    # ------------------------------------------------------------------ unmasked encoder (Models.py:869-894)
    def forward_encoder(self, x):
        """-> latent [N, T*9, D]: every token kept, natural order (the masked schedule with the full grid)."""
        T, L = self.input_size[0], self.input_size[1] ** 2
        N = x.shape[0]
        n1 = torch.arange(T, dtype=torch.float32).expand(N, T)          # increasing noise => ids_keep = identity
        n2 = torch.arange(L, dtype=torch.float32).expand(N, L)
        with torch.no_grad():
            _, _, _, st = self._run_forward(x, 0.0, (n1, n2), (T, L), want_latent=True, encoder_only=True)
        return st["latent"]

    def forward_mask_encoder(self, x, mask_ratio, noise=None, grid=None):
        """DualViT's name for the masked encoder (Models.py:896-921) = HSIMAE.forward_encoder; inference only."""
        return HSIMAE.forward_encoder(self, x, mask_ratio, noise, grid)

    def _packed_head(self, dev):
        w = self.cls_head.weight
        key = (w._version, self.cls_head.bias._version, dev)
        if self._head_pack is None or self._head_pack[0] != key:
            nc, k = w.shape
            npad = (nc + 15) // 16 * 16
            kp = (k + 31) // 32 * 32                          # K extent of the image: whole MFMA k-steps (T * D need not be one)
            img = torch.zeros(npad * kp, dtype=torch.bfloat16, device=dev)
            src = w.detach().to(device=dev, dtype=torch.float32).contiguous()
            desc = (_lib.PackDesc * 1)(_lib.PackDesc(src=src.data_ptr(), rows=nc, cols=k, transpose=0, n_off=0, k_off=0,
                                                     KS=kp // 32, dst=img.data_ptr()))
            table = torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8).clone().to(dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.load().hsimae_pack_matrix(table.data_ptr(), 1, src.numel(), stream), "hsimae_pack_matrix")
            bias = torch.zeros(npad, dtype=torch.float32, device=dev)      # padded: the F32 epilogue stores whole octets
            bias[:nc] = self.cls_head.bias.detach().to(device=dev, dtype=torch.float32)
            self._head_pack = (key, img, bias, npad, (src, table))
        return self._head_pack[1], self._head_pack[2], self._head_pack[3]

    def head(self, x, type="AGG"):
        """(class_pred [N, num_class], pooled [N, T*D]); 'AGG' only (the reference's default and only caller)."""
        if type != "AGG":
            raise NotImplementedError("only the 'AGG' head is built")
        N = x.shape[0]
        T, L, D = self.input_size[0], self.input_size[1] ** 2, self.dim
        lib = _lib.load()
        dev = x.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        x = x.contiguous()
        pooled = torch.empty(N, T * D, dtype=torch.float32, device=dev)
        _lib.check(lib.hsimae_agg_pool(x.data_ptr(), pooled.data_ptr(), N, T, L, D, stream), "hsimae_agg_pool")
        img, bias, npad = self._packed_head(dev)
        out = torch.empty(N, npad, dtype=torch.float32, device=dev)
        k, kp = T * D, (T * D + 31) // 32 * 32
        a = pooled
        if kp != k:                                           # zero-padded operand rows (whole k-steps)
            a = torch.zeros(N, kp, dtype=torch.float32, device=dev)
            a[:, :k] = pooled
        p = _lib.GemmParams(A=a.data_ptr(), lda=kp, M=N, N=npad, K=kp, n_valid=npad, W=img.data_ptr(),
                            bias=bias.data_ptr(), out=out.data_ptr(), ldo=npad)
        _lib.check(lib.hsimae_gemm(C.byref(p), _lib.A_F32, _lib.E_F32, stream), "hsimae_gemm")
        return out[:, :self.num_class], pooled

    # ------------------------------------------------------------------ stochastic depth (Models.py:235-263, 687-731)
    def drop_rates(self):
        """DropPath probability of every encoder block in execution order (blocks_1, blocks_2, blocks)."""
        dpr = [x.item() for x in torch.linspace(0, self.drop_path, self.depth)]
        out = []
        if self.s_depth > 0:
            out += dpr[:self.s_depth] + dpr[:self.s_depth]
        if self.s_depth < 12:
            out += dpr[self.s_depth:self.depth]
        return out

    def draw_drop_factors(self, N, len_t, len_l, device):
        """Per-sequence factors (0 or 1/keep) of one encoder pass, drawn in the reference's order: attention then MLP of
        every block; a block with rate 0 holds nn.Identity and draws nothing (Models.py:298)."""
        s, nf = self.s_depth, (self.depth - self.s_depth if self.s_depth < 12 else 0)
        nseq = ([N * len_t] * s + [N * len_l] * s if s > 0 else []) + [N] * nf
        out = []
        for p, n in zip(self.drop_rates(), nseq):
            if p == 0.0:
                out.append((None, None))
                continue
            keep = 1 - p
            pair = []
            for _ in range(2):
                m = torch.empty(n, 1, 1, device=device).bernoulli_(keep)
                if keep > 0.0:
                    m.div_(keep)
                pair.append(m.reshape(-1))
            out.append(tuple(pair))
        return out

    def _row_scales(self, factors, N, len_t, len_l, device):
        """[n_blocks, 2, N*K] fp32 per-row factors for hsimae_io.drop_scale, rows in the kernels' (n, t, l) order:
        a spatial block's sequence is (n, t) ('b (t l) c -> (b t) l c'), a spectral block's (n, l), a fusion block's n."""
        if factors is None or all(a is None and b is None for a, b in factors):
            return None
        s = self.s_depth
        rows = torch.ones(len(factors), 2, N, len_t, len_l, dtype=torch.float32, device=device)
        for e, pair in enumerate(factors):
            for j, m in enumerate(pair):
                if m is None:
                    continue
                m = m.to(device=device, dtype=torch.float32)
                if s > 0 and e < s:
                    rows[e, j] = m.view(N, len_t, 1)
                elif s > 0 and e < 2 * s:
                    rows[e, j] = m.view(N, 1, len_l)
                else:
                    rows[e, j] = m.view(N, 1, 1)
        return rows.reshape(len(factors), 2, N * len_t * len_l).contiguous()

    # ------------------------------------------------------------------ training step
    def _train_forward(self, imgs, imgs_u, mask_ratio, noise, grid, drop_factors):
        dev = imgs.device
        T, L = self.input_size[0], self.input_size[1] ** 2
        N = imgs.shape[0]
        use_drop = self.training and self.drop_path > 0.0
        f_cls = f_rec = None
        if drop_factors is not None:
            f_cls, f_rec = drop_factors
        elif use_drop:
            f_cls = self.draw_drop_factors(N, T, L, dev)              # forward_encoder's draws come first (Models.py:976)
        n1 = torch.arange(T, dtype=torch.float32).expand(N, T)
        n2 = torch.arange(L, dtype=torch.float32).expand(N, L)
        _, _, _, st_cls = self._run_forward(imgs, 0.0, (n1, n2), (T, L), want_latent=True, encoder_only=True,
                                            drop_scale=self._row_scales(f_cls, N, T, L, dev))
        class_pred, pooled = self.head(st_cls["latent"])
        saved = {"cls": st_cls, "pooled": pooled, "N": N, "rec": None}
        out = {"class_pred": class_pred, "saved": saved}
        if imgs_u is not None:
            imgs_all = torch.concat([imgs, imgs_u], dim=0)
            Na = imgs_all.shape[0]
            # RNG order of forward_mask_encoder: grid (python random), noise_1, noise_2, then the blocks' DropPaths
            len_t, len_l = grid if grid is not None else self.get_dim_patches(T, L, mask_ratio)
            if noise is None:
                noise = (torch.rand(Na, T, device=dev), torch.rand(Na, L, device=dev))
            if drop_factors is None and use_drop:
                f_rec = self.draw_drop_factors(Na, int(len_t), int(len_l), dev)
            loss_rec, pred_rec, mask, st_rec = self._run_forward(
                imgs_all, mask_ratio, noise, (int(len_t), int(len_l)), want_latent=False,
                drop_scale=self._row_scales(f_rec, Na, int(len_t), int(len_l), dev))
            saved["rec"] = st_rec
            out.update(loss_rec=loss_rec, pred_rec=pred_rec, mask=mask)
        return out

    def _train_backward(self, saved, g_loss, g_cls):
        lib, cfg = _lib.load(), self._config()
        dev = self._flat.device
        gw = gb = None
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            scratch = self._flat_scratch
            scratch.zero_()                                   # weight grads are accumulated with atomics
            nocb = _lib.BUCKET_CB(0)
            det = self._det_buffer(dev)
            for st in (saved["rec"], saved["cls"]):
                if st is not None:
                    st["io"].det_acc = det
            if saved["rec"] is not None and g_loss is not None:
                saved["rec"].check_alive()
                saved["rec"]["_done"] = True
                _lib.check(lib.hsimae_backward(C.byref(cfg), C.byref(saved["rec"]["io"]), scratch.data_ptr(), nocb, None, stream),
                           "hsimae_backward")
                scratch.mul_(g_loss)                          # chain rule with d/d(loss_rec) (lamda in the reference's loop)
            if g_cls is not None:
                # head (Models.py:962-970): class_pred = pooled W^T + b, pooled[n, t*D + c] = mean_l latent[n, t, l, c]
                N, T, L, D = saved["N"], self.input_size[0], self.input_size[1] ** 2, self.dim
                g_cls = g_cls.to(torch.float32).contiguous()
                w = self.cls_head.weight.detach().contiguous()
                gw, gb = torch.empty_like(w), torch.empty(w.shape[0], dtype=torch.float32, device=dev)
                dlat = torch.empty(N, T, L, D, dtype=torch.float32, device=dev)
                _lib.check(lib.hsimae_head_bwd(g_cls.data_ptr(), saved["pooled"].data_ptr(), w.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                               dlat.data_ptr(), N, w.shape[0], T, L, D, stream), "hsimae_head_bwd")
                saved["cls"].check_alive()
                saved["cls"]["_done"] = True
                _lib.check(lib.hsimae_encode_backward(C.byref(cfg), C.byref(saved["cls"]["io"]), dlat.data_ptr(),
                                                      scratch.data_ptr(), nocb, None, stream), "hsimae_encode_backward")
        self._apply_grads(scratch, 1.0)
        for st in (saved["rec"], saved["cls"]):
            if st is not None:
                st.release()
        return gw, gb

    def forward(self, imgs, imgs_u=None, mask_ratio=0.75, noise=None, grid=None, drop_factors=None):
        """-> class_pred, or (loss_rec, pred_rec, mask, class_pred) with imgs_u (Models.py:975-991).
        `noise`, `grid` as in HSIMAE.forward; `drop_factors=(classification pass, reconstruction pass)`, each a list of
        (attn, mlp) per-sequence factor vectors per encoder block (None entries = no DropPath), replaces the draws."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if imgs.is_cuda:
                self._ensure_flat(imgs.device)
            if self._reducer is not None:
                raise NotImplementedError("data-parallel fine-tuning is not built (pretraining only)")
            anchor = self._anchor if self._anchor is not None else torch.zeros((), requires_grad=True)
            return _FineTuneStep.apply(anchor, self.cls_head.weight, self.cls_head.bias, self, imgs, imgs_u, mask_ratio,
                                       noise, grid, drop_factors)
        with torch.no_grad():
            out = self._train_forward(imgs, imgs_u, mask_ratio, noise, grid, drop_factors)
        if imgs_u is None:
            return out["class_pred"]
        return out["loss_rec"], out["pred_rec"], out["mask"], out["class_pred"]


class HSIViT(DualViT):
    """The encoder-only classifier the reference evaluates fine-tuned checkpoints with (`Models.HSIViT`, Models.py:996-1167;
    `Model_Finetuning.test_model` :243-300): `forward(imgs) -> class logits`, 385-entry state_dict (encoder + `cls_head`),
    loaded key-filtered from a DualViT checkpoint.  Inference only.  The kernels' parameter layout always has a decoder, so a
    minimal one exists behind the scenes; it is not part of the module tree (state_dict / named_parameters do not list it).
    (Its random initialisation differs from the reference's stream: the class exists to load checkpoints.)"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0,
                 norm_layer=nn.LayerNorm, bands=16, b_patch_size=4, num_class=100, no_qkv_bias=False, trunc_init=False,
                 drop_rate=0., drop_path=0., s_depth=6, **kwargs):
        super().__init__(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, depth=depth,
                         s_depth=s_depth, num_heads=num_heads, mlp_ratio=mlp_ratio, norm_layer=norm_layer, bands=bands,
                         b_patch_size=b_patch_size, num_class=num_class, no_qkv_bias=no_qkv_bias, trunc_init=trunc_init,
                         drop_path=drop_path, decoder_embed_dim=32, decoder_depth=1, decoder_num_heads=4, norm_pix_loss=False)
        self._config()                                                  # cached while the decoder is still visible
        full = [p for n, p in self.named_parameters() if not n.startswith("cls_head.")]
        object.__setattr__(self, "_full_params", full)                  # flat-buffer order, decoder included
        hidden = {}
        for name in ("mask_token", "decoder_pos_embed"):
            hidden[name] = self._parameters.pop(name)
        for name in ("decoder_embed", "decoder_blocks", "decoder_norm", "decoder_pred"):
            hidden[name] = self._modules.pop(name)
        object.__setattr__(self, "_hidden_decoder", hidden)             # keeps them alive, outside the module tree

    def _plist(self):
        return self._full_params

    def _apply(self, fn, *a, **k):                                       # .to(device) must move the hidden decoder too
        out = super()._apply(fn, *a, **k)
        for v in self._hidden_decoder.values():
            if isinstance(v, nn.Module):
                v._apply(fn)
            else:
                v.data = fn(v.data)
        return out

    def forward(self, imgs):
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("hsimae_amd.HSIViT is the evaluation model (inference only); fine-tune with DualViT")
        with torch.no_grad():
            latent = self.forward_encoder(imgs)
            pred, _ = self.head(latent)
        return pred
