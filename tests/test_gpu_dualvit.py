"""Row N3 on the GPU: hsimae_amd.DualViT inference forward (hsimae_encode + hsimae_agg_pool + head GEMM) against the
fixture recorded from the reference DualViT and against the oracle at Base width."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hsimae_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
FX = np.load(os.path.join(ROOT, "tests", "golden", "dualvit_tiny.npz"))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def rms_rel(a, b):
    return float((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt())


def test_tiny_dualvit_against_reference_fixture():
    from hsimae_amd import DualViT
    m = quiet(DualViT, img_size=9, patch_size=3, in_chans=1, bands=32, b_patch_size=8, embed_dim=32, depth=3, s_depth=2,
              num_heads=2, num_class=11, trunc_init=True, drop_path=0.2, decoder_embed_dim=32, decoder_depth=2,
              decoder_num_heads=4, norm_pix_loss=True)
    m.load_state_dict({k[3:]: torch.from_numpy(FX[k]) for k in FX.files if k.startswith("sd/")})
    m = m.cuda().eval()
    x = torch.from_numpy(FX["x"]).cuda()
    lat = m.forward_encoder(x).cpu()
    assert rms_rel(lat, torch.from_numpy(FX["latent"])) < 5e-3          # bf16 MFMA operands, fp32 accumulation
    pred = m(x).cpu()
    ref = torch.from_numpy(FX["class_pred"])
    assert pred.shape == ref.shape and rms_rel(pred, ref) < 1e-2
    assert torch.equal(pred.argmax(1), ref.argmax(1))
    # dual-branch call: the masked path on concat(imgs, imgs_u) plus the same class_pred
    loss, rec, mask, pred2 = m(x, x.flip(0), mask_ratio=0.5)
    assert torch.isfinite(loss) and rec.shape == (12, 1, 32, 9, 9) and mask.shape == rec.shape and torch.equal(pred2.cpu(), pred)
    m.train()
    with pytest.raises(NotImplementedError):
        m(x)


def test_base_width_dualvit_against_oracle():
    from hsimae_amd import DualViT
    cfg = O.OracleConfig(bands=96)
    state = O.init_state(cfg, seed=4, std=0.02)
    g = torch.Generator().manual_seed(8)
    state["cls_head.weight"] = torch.randn(16, 128 * 12, generator=g) * 0.02
    state["cls_head.bias"] = torch.randn(16, generator=g) * 0.05
    m = quiet(DualViT, img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, num_class=16, embed_dim=128, depth=12,
              num_heads=8, s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True,
              trunc_init=True)
    m.load_state_dict(state)
    m = m.cuda().eval()
    x = torch.rand(24, 1, 96, 9, 9, generator=g)
    ref_pred, ref_pool = O.dualvit_classify(state, cfg, x)
    pred = m(x.cuda()).cpu()
    assert rms_rel(pred, ref_pred) < 1e-2, rms_rel(pred, ref_pred)
    lat = m.forward_encoder(x.cuda())
    _, pooled = m.head(lat)
    assert rms_rel(pooled.cpu(), ref_pool) < 5e-3
