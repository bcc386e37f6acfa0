"""Row N3 on CPU: the oracle's unmasked-encoder / AGG-head restatement against the fixture recorded from the reference
DualViT (tests/golden/make_golden_dualvit.py), and the DualViT mirror's state_dict layout against the manifest."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hsimae_oracle as O  # noqa: E402

FX = np.load(os.path.join(ROOT, "tests", "golden", "dualvit_tiny.npz"))
CFG = dict(bands=32, embed_dim=32, depth=3, s_depth=2, num_heads=2, decoder_embed_dim=32, decoder_depth=2, decoder_num_heads=4)


def tiny_state():
    return {k[3:]: torch.from_numpy(FX[k]) for k in FX.files if k.startswith("sd/")}


def test_oracle_unmasked_encoder_and_agg_head_match_reference_dualvit():
    cfg = O.OracleConfig(**CFG)
    P = tiny_state()
    x = torch.from_numpy(FX["x"])
    lat = O.encode_unmasked(P, cfg, x)
    assert float((lat - torch.from_numpy(FX["latent"])).abs().max()) < 2e-5
    pred, pooled = O.dualvit_classify(P, cfg, x)
    assert float((pooled - torch.from_numpy(FX["pooled"])).abs().max()) < 2e-5
    assert float((pred - torch.from_numpy(FX["class_pred"])).abs().max()) < 2e-5


def test_dualvit_mirror_state_dict_matches_reference_manifest():
    import contextlib
    import io
    from hsimae_amd import DualViT
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["DualViT_base32"]
    with contextlib.redirect_stdout(io.StringIO()):
        m = DualViT(img_size=9, patch_size=3, in_chans=1, bands=32, b_patch_size=8, num_class=16, embed_dim=128, depth=12,
                    num_heads=8, s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True,
                    trunc_init=True, drop_path=0.2)
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
    assert got == man                                   # 537 keys, same order (cls_head between norm and decoder_embed)
    assert len(m._plist()) == 535                       # the kernel library's flat layout is the autoencoder's
    # tiny fixture loads by name
    with contextlib.redirect_stdout(io.StringIO()):
        t = DualViT(img_size=9, patch_size=3, in_chans=1, num_class=11, trunc_init=True, drop_path=0.2, norm_pix_loss=True,
                    b_patch_size=8, **{k: v for k, v in CFG.items()})
    t.load_state_dict(tiny_state())
