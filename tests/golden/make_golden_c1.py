#!/usr/bin/env python3
"""Config-1 fixtures recorded from the REFERENCE (BASELINE.json configs[0]: HSIMAE-Base, 9x9x48 cubes, batch 64, mask 0.75,
PyTorch CPU fp32).  Runs only in the build container (imports /root/reference/Models.py in place; nothing of it is copied).

    python tests/golden/make_golden_c1.py

  c1_refscale.json/.npz   one fwd+bwd of the reference right after construction (`torch.manual_seed(0)`, trunc_init: the
                          reference's own weight scale, Models.py:439-459): loss, stage checksums, 532 gradient L2 norms,
                          the replayed noise and grid.  The GPU test rebuilds the weights from the seed through
                          hsimae_amd.HSIMAE (init RNG order is pinned by init_checksums.json).
  c1_trajectory.json/.npz F6 at config 1: 10 AdamW steps (the reference loop's optimizer, Model_Pretraining.py:80-86, lr 5e-3,
                          wd 5e-2, betas .9/.95, no scheduler) from the seed-3 construction: losses, grids, noise per step.
"""
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import make_mae, replay, stats, taps_forward  # noqa: E402  (helpers of the main generator; they import the reference)


def refscale():
    torch.manual_seed(0); random.seed(0)
    model = make_mae(48, 128)
    N = 64
    torch.manual_seed(1234)
    x = torch.rand(N, 1, 48, 9, 9)
    torch.manual_seed(99); random.seed(0)
    (loss, pred, mask), taps, n1, n2, lt, ll, cands = taps_forward(model, x, 0.75)
    loss.backward()
    out = {"N": N, "bands": 48, "len_t": lt, "len_l": ll, "candidates": [list(c) for c in cands],
           "loss_fp32": float(loss.item()), "model_seed": 0, "x_seed": 1234,
           "taps": {k: stats(v) for k, v in taps.items() if v.dtype != torch.int64},
           "pred_img": stats(pred), "mask_img_sum": float(mask.sum()),
           "grad_l2": {k: float(p.grad.double().norm()) for k, p in model.named_parameters() if p.grad is not None}}
    json.dump(out, open(os.path.join(HERE, "c1_refscale.json"), "w"))
    g = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    np.savez_compressed(os.path.join(HERE, "c1_refscale.npz"), noise_1=n1.numpy(), noise_2=n2.numpy(),
                        ids_keep=taps["ids_keep"].numpy().astype(np.int16),
                        latent=taps["latent"].numpy().astype(np.float32)[:4],
                        pred=taps["pred"].numpy().astype(np.float32)[:2],
                        g_blocks0_w2=g["blocks.0.mlp.w2.weight"].numpy(), g_b1_0_q=g["blocks_1.0.attn.q.weight"].numpy(),
                        g_dec7_w1=g["decoder_blocks.7.mlp.w1.weight"].numpy(), g_pe=g["patch_embed.proj.weight"].numpy())
    print("c1_refscale: loss", out["loss_fp32"])


def trajectory():
    torch.manual_seed(3); random.seed(3)
    model = make_mae(48, 128)
    N = 64
    torch.manual_seed(4321)
    x = torch.rand(N, 1, 48, 9, 9)
    nd = ["bias", "norm"]
    groups = [{"params": [p for n, p in model.named_parameters() if not any(k in n for k in nd)], "weight_decay": 5e-2},
              {"params": [p for n, p in model.named_parameters() if any(k in n for k in nd)], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=5e-3, weight_decay=5e-2, betas=(0.9, 0.95))
    torch.manual_seed(55); random.seed(55)
    losses, grids, arrs = [], [], {}
    for step in range(10):
        (loss, _, _), n1, n2, lt, ll, _ = replay(model, x, 0.75)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss.item())); grids.append([lt, ll])
        arrs[f"n1_{step}"] = n1.numpy(); arrs[f"n2_{step}"] = n2.numpy()
    np.savez_compressed(os.path.join(HERE, "c1_trajectory.npz"), **arrs)
    json.dump({"losses": losses, "grids": grids, "lr": 5e-3, "wd": 5e-2, "betas": [0.9, 0.95], "ratio": 0.75,
               "model_seed": 3, "x_seed": 4321, "N": N, "bands": 48}, open(os.path.join(HERE, "c1_trajectory.json"), "w"))
    print("c1_trajectory:", losses)


if __name__ == "__main__":
    torch.set_num_threads(8)
    refscale()
    trajectory()
