#!/usr/bin/env python3
"""Golden fixture for row N3 (training) from the REFERENCE `Models.DualViT` in train mode: one fine-tuning step
(Model_Finetuning.py:150-156) of the small model of make_golden_dualvit.py — forward(imgs, imgs_u, mask_ratio) with
DropPath 0.2, loss = lamda * loss_rec + CrossEntropy(ignore_index=0), backward.  The random draws of the step
(DropPath factors of both encoder passes, grid choice, masking noise) are replayed from the same seeds and stored, and
the oracle restatement is checked against the reference before anything is written.  Build container only.

    python tests/golden/make_golden_dualvit_train.py      ->  tests/golden/dualvit_train_tiny.npz
"""
import contextlib
import io
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
with contextlib.redirect_stdout(io.StringIO()):
    import Models as R  # noqa: E402
from oracle import hsimae_oracle as O  # noqa: E402

MASK_RATIO, LAMDA, DROP = 0.5, 5.0, 0.2


def build():
    torch.manual_seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        m = R.DualViT(img_size=9, patch_size=3, in_chans=1, bands=32, b_patch_size=8, embed_dim=32, depth=3, s_depth=2,
                      num_heads=2, num_class=11, trunc_init=True, drop_path=DROP, decoder_embed_dim=32, decoder_depth=2,
                      decoder_num_heads=4, norm_pix_loss=True)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
    return m


def main():
    m = build()
    base = np.load(os.path.join(HERE, "dualvit_tiny.npz"))
    for k, v in m.state_dict().items():           # same weights as the inference fixture: stored once, there
        assert np.array_equal(base["sd/" + k], v.numpy()), k
    g = torch.Generator().manual_seed(17)
    x = torch.rand(4, 1, 32, 9, 9, generator=g)
    x_u = torch.rand(6, 1, 32, 9, 9, generator=g)
    y = torch.tensor([3, 0, 7, 10])                # class 0 = unlabeled (ignore_index, Model_Finetuning.py:108)
    cfg = O.OracleConfig(bands=32, embed_dim=32, depth=3, s_depth=2, num_heads=2, decoder_embed_dim=32, decoder_depth=2,
                         decoder_num_heads=4, norm_pix_loss=True)

    # ---- the reference step
    m.train()
    random.seed(3); torch.manual_seed(21)
    loss_rec, _, _, outputs = m(x, x_u, mask_ratio=MASK_RATIO)
    loss = LAMDA * loss_rec + torch.nn.functional.cross_entropy(outputs, y, reduction="mean", ignore_index=0)
    loss.backward()
    len_t, len_l = int(m.len_t), int(m.len_l)

    # ---- replay of its random draws, in its order (forward_encoder's DropPaths; get_dim_patches; noise_1; noise_2;
    #      forward_mask_encoder's DropPaths)
    random.seed(3); torch.manual_seed(21)
    drops_cls = O.draw_drop_factors(cfg, DROP, 4, cfg.T, cfg.L)
    grid = O.choose_grid(cfg.T, cfg.L, MASK_RATIO, random)
    assert tuple(grid) == (len_t, len_l)
    n1 = torch.rand(10, cfg.T); n2 = torch.rand(10, cfg.L)
    drops_rec = O.draw_drop_factors(cfg, DROP, 10, len_t, len_l)

    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    o_rec, o_pred, o_loss, o_grads = O.dualvit_train_step(P, cfg, x, x_u, y, LAMDA, n1.numpy(), n2.numpy(), len_t, len_l,
                                                           drops_cls, drops_rec)
    assert abs(float(o_rec) - float(loss_rec)) <= 1e-6 * abs(float(loss_rec)), (float(o_rec), float(loss_rec))
    assert torch.allclose(o_pred, outputs.detach(), rtol=1e-5, atol=1e-6)
    worst = 0.0
    for n, p in m.named_parameters():
        if p.grad is None:
            assert n not in o_grads or float(o_grads[n].abs().max()) == 0.0, n
            continue
        d = float((o_grads[n] - p.grad).abs().max()) / (float(p.grad.abs().max()) + 1e-12)
        worst = max(worst, d)
    assert worst < 2e-4, worst
    print("oracle vs reference: loss_rec", float(o_rec), float(loss_rec), "worst grad rel", worst)

    out = {"x": x.numpy(), "x_u": x_u.numpy(), "y": y.numpy(), "noise_1": n1.numpy(), "noise_2": n2.numpy(),
           "grid": np.array([len_t, len_l]), "mask_ratio": np.float32(MASK_RATIO), "lamda": np.float32(LAMDA),
           "drop_path": np.float32(DROP), "loss_rec": loss_rec.detach().numpy(), "class_pred": outputs.detach().numpy(),
           "loss": loss.detach().numpy()}
    for tag, drops in (("cls", drops_cls), ("rec", drops_rec)):
        for e, (a, b) in enumerate(drops):
            if a is not None:
                out[f"drop_{tag}/{e}/attn"] = a.numpy(); out[f"drop_{tag}/{e}/mlp"] = b.numpy()
    for n, p in m.named_parameters():
        if p.grad is not None:
            out["grad/" + n] = p.grad.numpy()
    path = os.path.join(HERE, "dualvit_train_tiny.npz")
    np.savez_compressed(path, **out)
    print("grid", grid, "grads", sum(k.startswith("grad/") for k in out), "bytes", os.path.getsize(path))


if __name__ == "__main__":
    main()
