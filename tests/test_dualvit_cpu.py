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


# ------------------------------------------------------------------ training step (DropPath, both branches, backward)
FXT = np.load(os.path.join(ROOT, "tests", "golden", "dualvit_train_tiny.npz"))


def fixture_drops(tag, n_blocks=5):
    out = []
    for e in range(n_blocks):
        k = f"drop_{tag}/{e}/attn"
        out.append((torch.from_numpy(FXT[k]), torch.from_numpy(FXT[f"drop_{tag}/{e}/mlp"])) if k in FXT.files else (None, None))
    return out


def test_oracle_finetune_step_matches_reference_fixture():
    """loss_rec, class_pred and every parameter gradient of one reference fine-tuning step (train mode, DropPath 0.2,
    lamda * loss_rec + CE(ignore_index=0)) from the recorded draws."""
    cfg = O.OracleConfig(norm_pix_loss=True, **CFG)
    P = tiny_state()
    rec, pred, loss, grads = O.dualvit_train_step(
        P, cfg, torch.from_numpy(FXT["x"]), torch.from_numpy(FXT["x_u"]), torch.from_numpy(FXT["y"]), float(FXT["lamda"]),
        FXT["noise_1"], FXT["noise_2"], int(FXT["grid"][0]), int(FXT["grid"][1]), fixture_drops("cls"), fixture_drops("rec"))
    assert abs(float(rec) - float(FXT["loss_rec"])) < 1e-6
    assert abs(float(loss) - float(FXT["loss"])) < 1e-5
    assert float((pred - torch.from_numpy(FXT["class_pred"])).abs().max()) < 1e-5
    names = [k[5:] for k in FXT.files if k.startswith("grad/")]
    assert len(names) == 138
    for n in names:
        ref = torch.from_numpy(FXT["grad/" + n])
        assert float((grads[n] - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-9, n


def test_drop_rates_and_draw_order_match_reference_layout():
    cfg = O.OracleConfig(norm_pix_loss=True, **CFG)
    rates = O.drop_rates(cfg, 0.2)                       # depth 3, s_depth 2: blocks_1[0,1], blocks_2[0,1], blocks[0]
    assert np.allclose(rates, [0.0, 0.1, 0.0, 0.1, 0.2])
    assert [a is None for a, _ in fixture_drops("cls")] == [True, False, True, False, False]
    torch.manual_seed(21)
    d = O.draw_drop_factors(cfg, 0.2, 4, cfg.T, cfg.L)
    for e, (a, b) in enumerate(fixture_drops("cls")):    # the first draws after the seed are the classification pass's
        if a is not None:
            assert torch.equal(d[e][0], a) and torch.equal(d[e][1], b)
    from hsimae_amd import DualViT
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        m = DualViT(img_size=9, patch_size=3, in_chans=1, bands=32, b_patch_size=8, embed_dim=32, depth=3, s_depth=2,
                    num_heads=2, num_class=11, trunc_init=True, drop_path=0.2, decoder_embed_dim=32, decoder_depth=2,
                    decoder_num_heads=4, norm_pix_loss=True)
    assert np.allclose(m.drop_rates(), rates)
    torch.manual_seed(21)
    dm = m.draw_drop_factors(4, cfg.T, cfg.L, torch.device("cpu"))     # same stream as the reference on CPU
    for e, (a, b) in enumerate(d):
        assert (a is None) == (dm[e][0] is None)
        if a is not None:
            assert torch.equal(dm[e][0], a) and torch.equal(dm[e][1], b)
    # per-row expansion: a spatial block's sequence is (n, t), a spectral block's (n, l), a fusion block's n
    N, t, l = 4, cfg.T, cfg.L
    rows = m._row_scales(dm, N, t, l, torch.device("cpu")).view(5, 2, N, t, l)
    assert torch.all(rows[0] == 1) and torch.all(rows[2] == 1)
    assert torch.equal(rows[1, 0], dm[1][0].view(N, t, 1).expand(N, t, l))
    assert torch.equal(rows[3, 1], dm[3][1].view(N, 1, l).expand(N, t, l))
    assert torch.equal(rows[4, 0], dm[4][0].view(N, 1, 1).expand(N, t, l))


def test_finetune_host_helpers_split_and_scores():
    """spilt_dataset (Utils/Preprocessing.py:276-300) and the OA / AA / kappa restatement against sklearn."""
    from sklearn import metrics
    from hsimae_amd.finetune_train import scores, spilt_dataset
    np.random.seed(3)
    label = np.array([1, 2, 3] * 10 + [1] * 6)
    data = list(range(len(label)))
    tr, tr_y, va, va_y = spilt_dataset(data, label, training_ratio=0.5)
    assert sorted(tr + va) == data and len(va) == 8 + 5 + 5
    assert all(label[i] == y for i, y in zip(tr, tr_y)) and all(label[i] == y for i, y in zip(va, va_y))
    rng = np.random.default_rng(1)
    gt = rng.integers(0, 5, 400)
    pred = np.where(rng.random(400) < 0.7, gt, rng.integers(1, 5, 400))
    oa, aa, kappa, ca = scores(gt, pred)
    g, p = gt[gt != 0] - 1, pred[gt != 0] - 1
    assert abs(oa - metrics.accuracy_score(g, p)) < 1e-12
    assert abs(aa - np.mean(metrics.recall_score(g, p, average=None, labels=np.unique(g)))) < 1e-12
    assert abs(kappa - metrics.cohen_kappa_score(g, p)) < 1e-12


def test_hsivit_mirror_state_dict_matches_reference_manifest():
    """hsimae_amd.HSIViT (the evaluation model of Model_Finetuning.test_model) exposes the reference HSIViT's 385 keys /
    shapes, and a DualViT state_dict loads into it key-filtered the way the reference does (:253-261)."""
    import contextlib
    import io
    from hsimae_amd import DualViT, HSIViT
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["HSIViT_base32"]
    with contextlib.redirect_stdout(io.StringIO()):
        v = HSIViT(img_size=9, patch_size=3, in_chans=1, bands=32, b_patch_size=8, num_class=16, embed_dim=128, depth=12,
                   num_heads=8, s_depth=9, trunc_init=True)
        d = DualViT(img_size=9, patch_size=3, in_chans=1, bands=32, b_patch_size=8, num_class=16, embed_dim=128, depth=12,
                    num_heads=8, s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, trunc_init=True)
    sd = v.state_dict()
    assert [(k, list(t.shape), str(t.dtype).replace("torch.", "")) for k, t in sd.items()] == [tuple(e) if False else (e[0], e[1], e[2]) for e in man]
    assert [n for n, _ in v.named_parameters()] == [e[0] for e in man]
    model_dict = v.state_dict()
    model_dict.update({k: t for k, t in d.state_dict().items() if k in model_dict})
    v.load_state_dict(model_dict)
    assert torch.equal(v.blocks_1[3].mlp.w1.weight, d.blocks_1[3].mlp.w1.weight) and torch.equal(v.cls_head.bias, d.cls_head.bias)
