"""Row N3: the dual-branch fine-tuning model's FORWARD (`Models.DualViT`, Models.py:637-993) on the
MI355X-native kernels — inference only.

  class_pred = cls_head(AGG(norm(encoder(imgs))))                      (forward_encoder :869-894, head :962-970)
  with imgs_u: the masked-autoencoder path on concat(imgs, imgs_u)      (forward :975-993) — the pretraining hot path

Same constructor keywords, same parameter names / order / init stream as the reference (the head is registered
between `norm` and `decoder_embed`), so fine-tuned checkpoints load both ways.  Everything runs through the HIP
library: `hsimae_encode` (unmasked encoder = the masked schedule with the full token grid and identity order),
`hsimae_agg_pool`, `hsimae_gemm` for the head.  NOT built: the fine-tuning BACKWARD (classification loss through
the head and the unmasked encoder, DropPath); calling the model in training mode raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .model import HSIMAE


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
        self.drop_path = drop_path            # stochastic depth is a training-time op: identity in this (eval) forward
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

    def _packed_head(self, dev):
        w = self.cls_head.weight
        key = (w._version, self.cls_head.bias._version, dev)
        if self._head_pack is None or self._head_pack[0] != key:
            nc, k = w.shape
            npad = (nc + 15) // 16 * 16
            img = torch.zeros(npad * k, dtype=torch.bfloat16, device=dev)
            src = w.detach().to(device=dev, dtype=torch.float32).contiguous()
            desc = (_lib.PackDesc * 1)(_lib.PackDesc(src=src.data_ptr(), rows=nc, cols=k, transpose=0, n_off=0, k_off=0,
                                                     KS=k // 32, dst=img.data_ptr()))
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
        p = _lib.GemmParams(A=pooled.data_ptr(), lda=T * D, M=N, N=npad, K=T * D, n_valid=npad, W=img.data_ptr(),
                            bias=bias.data_ptr(), out=out.data_ptr(), ldo=npad)
        _lib.check(lib.hsimae_gemm(C.byref(p), _lib.A_F32, _lib.E_F32, stream), "hsimae_gemm")
        return out[:, :self.num_class], pooled

    def forward(self, imgs, imgs_u=None, mask_ratio=0.75, noise=None, grid=None):
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("hsimae_amd.DualViT is the inference forward (row N3); fine-tuning backward / "
                                      "DropPath are not built — call .eval() or wrap in torch.no_grad()")
        latent = self.forward_encoder(imgs)
        class_pred, _ = self.head(latent)
        if imgs_u is None:
            return class_pred
        imgs_all = torch.concat([imgs, imgs_u], dim=0)
        with torch.no_grad():
            loss_rec, pred_rec, mask, _ = self._run_forward(imgs_all, mask_ratio, noise, grid, want_latent=False)
        return loss_rec, pred_rec, mask, class_pred
