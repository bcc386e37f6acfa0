#!/usr/bin/env python3
"""Golden fixture for row N3 from the REFERENCE `Models.DualViT` (eval mode): a small model's state_dict, inputs and the
classification output of `forward(imgs)`, plus the unmasked encoder latent.  Runs only in the build container.

    python tests/golden/make_golden_dualvit.py      ->  tests/golden/dualvit_tiny.npz
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
with contextlib.redirect_stdout(io.StringIO()):
    import Models as R  # noqa: E402


def main():
    torch.manual_seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        m = R.DualViT(img_size=9, patch_size=3, in_chans=1, bands=32, b_patch_size=8, embed_dim=32, depth=3, s_depth=2,
                      num_heads=2, num_class=11, trunc_init=True, drop_path=0.2, decoder_embed_dim=32, decoder_depth=2,
                      decoder_num_heads=4, norm_pix_loss=True)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():                       # non-degenerate LayerNorm / bias values (init leaves them at 1 / 0)
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
    m.eval()
    x = torch.rand(6, 1, 32, 9, 9, generator=g)
    with torch.no_grad():
        latent = m.forward_encoder(x)
        pred = m(x)
        pred2, pooled = m.head(latent)
    assert torch.equal(pred, pred2)
    out = {"x": x.numpy(), "latent": latent.numpy(), "class_pred": pred.numpy(), "pooled": pooled.numpy()}
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "dualvit_tiny.npz"), **out)
    print("keys", len(m.state_dict()), "pred", tuple(pred.shape), "bytes", os.path.getsize(os.path.join(HERE, "dualvit_tiny.npz")))


if __name__ == "__main__":
    main()
