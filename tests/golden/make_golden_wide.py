#!/usr/bin/env python3
"""Fixtures at the WIDE configurations recorded from the REFERENCE (BASELINE.json configs[2] and configs[4] at small batch:
HSIMAE-Large = 9x9x96 cubes, embed 256 / 16 heads; the D = 512 / 32-head / 9x9x192 model this repo calls Huge).  PyTorch CPU
fp32, the reference's own weights right after construction.  Runs only in the build container (imports
/root/reference/Models.py in place; nothing of it is copied).

    python tests/golden/make_golden_wide.py

  c3_refscale.json/.npz   Large, N = 16: loss, prediction-image checksums, 532 gradient L2 norms, replayed noise / grid, ids_keep,
                          two full gradient tensors and the first 64 rows of two wide ones
  c5_refscale.json/.npz   D = 512, N = 4: the same
"""
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import make_mae, stats, taps_forward  # noqa: E402  (helpers of the main generator; they import the reference)


def refscale(tag, bands, dim, heads, N, model_seed, x_seed, rng_seed):
    torch.manual_seed(model_seed); random.seed(model_seed)
    model = make_mae(bands, dim, heads=heads)
    torch.manual_seed(x_seed)
    x = torch.rand(N, 1, bands, 9, 9)
    torch.manual_seed(rng_seed); random.seed(rng_seed)
    (loss, pred, mask), taps, n1, n2, lt, ll, cands = taps_forward(model, x, 0.75)
    loss.backward()
    out = {"N": N, "bands": bands, "dim": dim, "heads": heads, "len_t": lt, "len_l": ll, "loss_fp32": float(loss.item()),
           "model_seed": model_seed, "x_seed": x_seed, "pred_img": stats(pred), "mask_img_sum": float(mask.sum()),
           "grad_l2": {k: float(p.grad.double().norm()) for k, p in model.named_parameters() if p.grad is not None}}
    json.dump(out, open(os.path.join(HERE, f"{tag}_refscale.json"), "w"))
    g = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    np.savez_compressed(os.path.join(HERE, f"{tag}_refscale.npz"), noise_1=n1.numpy(), noise_2=n2.numpy(),
                        ids_keep=taps["ids_keep"].numpy().astype(np.int16),
                        latent=taps["latent"].numpy().astype(np.float32)[:2],
                        g_blocks0_w2=g["blocks.0.mlp.w2.weight"].numpy()[:64],        # first 64 rows (fixture size)
                        g_b1_0_q=g["blocks_1.0.attn.q.weight"].numpy()[:64],
                        g_dec7_w1=g["decoder_blocks.7.mlp.w1.weight"].numpy(), g_pe=g["patch_embed.proj.weight"].numpy())
    print(tag, "loss", out["loss_fp32"], "grid", lt, ll)


if __name__ == "__main__":
    torch.set_num_threads(8)
    refscale("c3", 96, 256, 16, 16, 7, 77, 707)
    refscale("c5", 192, 512, 32, 4, 9, 99, 909)
