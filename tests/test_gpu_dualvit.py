"""Row N3 on the GPU: hsimae_amd.DualViT — inference forward (hsimae_encode + hsimae_agg_pool + head GEMM) and the
fine-tuning step (DropPath, both branches, hsimae_encode_backward + hsimae_backward) — against the fixtures recorded
from the reference DualViT and against the oracle at Base width."""
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


# ------------------------------------------------------------------ fine-tuning step
FXT = np.load(os.path.join(ROOT, "tests", "golden", "dualvit_train_tiny.npz"))


def fixture_drops(tag, n_blocks=5):
    out = []
    for e in range(n_blocks):
        k = f"drop_{tag}/{e}/attn"
        out.append((torch.from_numpy(FXT[k]), torch.from_numpy(FXT[f"drop_{tag}/{e}/mlp"])) if k in FXT.files else (None, None))
    return out


def grad_report(model, ref_grads):
    """-> (worst RMS-relative error, its name) over every parameter with a reference gradient."""
    worst, wname = 0.0, None
    for n, p in model.named_parameters():
        if n not in ref_grads or n.endswith("attn.k.bias"):     # k.bias: the true gradient is exactly zero (softmax shift invariance)
            continue
        assert p.grad is not None, n
        r = ref_grads[n]
        scale = float(r.double().pow(2).mean().sqrt())
        if scale < 1e-9:
            continue
        e = float((p.grad.detach().cpu().double() - r.double()).pow(2).mean().sqrt()) / scale
        if e > worst:
            worst, wname = e, n
    return worst, wname


def test_tiny_finetune_step_against_reference_fixture():
    """Reference fine-tuning step (train mode, DropPath 0.2, lamda * loss_rec + CE(ignore_index=0), backward) replayed
    with the recorded draws: losses, logits and all 138 parameter gradients."""
    from hsimae_amd import DualViT
    m = quiet(DualViT, img_size=9, patch_size=3, in_chans=1, bands=32, b_patch_size=8, embed_dim=32, depth=3, s_depth=2,
              num_heads=2, num_class=11, trunc_init=True, drop_path=0.2, decoder_embed_dim=32, decoder_depth=2,
              decoder_num_heads=4, norm_pix_loss=True)
    m.load_state_dict({k[3:]: torch.from_numpy(FX[k]) for k in FX.files if k.startswith("sd/")})
    m = m.cuda().train()
    x, xu, y = (torch.from_numpy(FXT[k]).cuda() for k in ("x", "x_u", "y"))
    lam = float(FXT["lamda"])
    loss_rec, rec, mask, pred = m(x, xu, mask_ratio=float(FXT["mask_ratio"]),
                                  noise=(torch.from_numpy(FXT["noise_1"]), torch.from_numpy(FXT["noise_2"])),
                                  grid=(int(FXT["grid"][0]), int(FXT["grid"][1])),
                                  drop_factors=(fixture_drops("cls"), fixture_drops("rec")))
    loss = lam * loss_rec + torch.nn.functional.cross_entropy(pred, y, reduction="mean", ignore_index=0)
    loss.backward()
    assert abs(float(loss_rec) - float(FXT["loss_rec"])) < 1e-4 * float(FXT["loss_rec"])       # north-star gate
    assert rms_rel(pred.detach().cpu(), torch.from_numpy(FXT["class_pred"])) < 1e-2
    assert abs(float(loss) - float(FXT["loss"])) < 2e-3 * abs(float(FXT["loss"]))
    ref = {k[5:]: torch.from_numpy(FXT[k]) for k in FXT.files if k.startswith("grad/")}
    worst, wname = grad_report(m, ref)
    print(f"[finetune tiny] loss_rec {float(loss_rec):.6f} ref {float(FXT['loss_rec']):.6f}  worst grad rms-rel {worst:.2e} ({wname})")
    assert worst < 5e-2, (worst, wname)         # width 32: same bound as the tiny pretraining fixture (test_gpu_e2e.py)
    # a dropped sequence contributes nothing: with every factor 0 the blocks are the identity and the encoder's
    # block parameters get exactly zero gradient from the classification branch
    m.zero_grad(set_to_none=True)
    zeros = [(torch.zeros(n), torch.zeros(n)) for n in (4 * 4, 4 * 4, 4 * 9, 4 * 9, 4)]
    pred0 = m(x, drop_factors=(zeros, None))
    pred0.sum().backward()
    assert float(m.blocks[0].mlp.w1.weight.grad.abs().max()) == 0.0
    assert float(m.blocks_1[1].attn.q.weight.grad.abs().max()) == 0.0
    assert float(m.patch_embed.proj.weight.grad.abs().max()) > 0.0


def test_base_width_finetune_step_against_oracle():
    """Base width (the fused d = 128 kernels carry the DropPath factors): one step against the oracle."""
    from hsimae_amd import DualViT
    cfg = O.OracleConfig(bands=96, norm_pix_loss=True)
    state = O.init_state(cfg, seed=4, std=0.06)
    g = torch.Generator().manual_seed(8)
    state["cls_head.weight"] = torch.randn(16, 128 * 12, generator=g) * 0.02
    state["cls_head.bias"] = torch.randn(16, generator=g) * 0.05
    m = quiet(DualViT, img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, num_class=16, embed_dim=128, depth=12,
              num_heads=8, s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True,
              trunc_init=True, drop_path=0.2)
    m.load_state_dict(state)
    m = m.cuda().train()
    N, Nu = 6, 10
    x = torch.rand(N, 1, 96, 9, 9, generator=g)
    xu = torch.rand(Nu, 1, 96, 9, 9, generator=g)
    y = torch.tensor([1, 0, 5, 15, 3, 9])
    n1 = torch.rand(N + Nu, cfg.T, generator=g)
    n2 = torch.rand(N + Nu, cfg.L, generator=g)
    len_t, len_l = 6, 9                                   # mask ratio 0.5 at T = 12: (6, 9) or (9, 6)
    torch.manual_seed(33)
    d_cls = O.draw_drop_factors(cfg, 0.2, N, cfg.T, cfg.L)
    d_rec = O.draw_drop_factors(cfg, 0.2, N + Nu, len_t, len_l)
    lam = 5.0
    o_rec, o_pred, o_loss, o_grads = O.dualvit_train_step(state, cfg, x, xu, y, lam, n1.numpy(), n2.numpy(), len_t, len_l,
                                                           d_cls, d_rec)
    loss_rec, _, _, pred = m(x.cuda(), xu.cuda(), mask_ratio=0.5, noise=(n1, n2), grid=(len_t, len_l), drop_factors=(d_cls, d_rec))
    loss = lam * loss_rec + torch.nn.functional.cross_entropy(pred, y.cuda(), reduction="mean", ignore_index=0)
    loss.backward()
    assert abs(float(loss_rec) - float(o_rec)) < 1e-4 * float(o_rec), (float(loss_rec), float(o_rec))
    assert rms_rel(pred.detach().cpu(), o_pred) < 1e-2
    worst, wname = grad_report(m, o_grads)
    print(f"[finetune base] loss_rec {float(loss_rec):.6f} oracle {float(o_rec):.6f}  worst grad rms-rel {worst:.2e} ({wname})")
    assert worst < 3e-2, (worst, wname)
    # drawn (not injected) factors: the step runs and DropPath really is active in training mode
    m.zero_grad(set_to_none=True)
    torch.manual_seed(1)
    a = m(x.cuda())
    torch.manual_seed(2)
    b = m(x.cuda())
    assert not torch.equal(a, b)
    m.eval()
    with torch.no_grad():
        assert torch.equal(m(x.cuda()), m(x.cuda()))


def test_finetune_step_at_the_reference_scripts_default_widths_144_72():
    """Model_Finetuning.py:66-67 defaults (dim 144 = 9 heads of 16, dec_dim 72 = 9 heads of 8, GWPCA's 32 bands): widths that
    are not multiples of 32 run zero-padded to 160 / 96; one training step against the oracle."""
    from hsimae_amd import DualViT
    cfg = O.OracleConfig(bands=32, embed_dim=144, num_heads=9, depth=12, s_depth=6, decoder_embed_dim=72, decoder_depth=2,
                         decoder_num_heads=9, norm_pix_loss=True)
    state = O.init_state(cfg, seed=2, std=0.05)
    g = torch.Generator().manual_seed(18)
    state["cls_head.weight"] = torch.randn(16, 144 * 4, generator=g) * 0.02
    state["cls_head.bias"] = torch.randn(16, generator=g) * 0.05
    m = quiet(DualViT, img_size=9, patch_size=3, in_chans=1, bands=32, b_patch_size=8, num_class=16, embed_dim=144, depth=12,
              num_heads=9, s_depth=6, decoder_embed_dim=72, decoder_depth=2, decoder_num_heads=9, norm_pix_loss=True,
              trunc_init=True, drop_path=0.2)
    m.load_state_dict(state)
    m = m.cuda().train()
    N, Nu = 5, 7
    x = torch.rand(N, 1, 32, 9, 9, generator=g)
    xu = torch.rand(Nu, 1, 32, 9, 9, generator=g)
    y = torch.tensor([1, 0, 5, 15, 3])
    n1 = torch.rand(N + Nu, cfg.T, generator=g)
    n2 = torch.rand(N + Nu, cfg.L, generator=g)
    len_t, len_l = 2, 4                                   # mask ratio 0.8 at T = 4: (2, 4) or (4, 2)
    torch.manual_seed(3)
    d_cls = O.draw_drop_factors(cfg, 0.2, N, cfg.T, cfg.L)
    d_rec = O.draw_drop_factors(cfg, 0.2, N + Nu, len_t, len_l)
    lam = 1.0
    o_rec, o_pred, o_loss, o_grads = O.dualvit_train_step(state, cfg, x, xu, y, lam, n1.numpy(), n2.numpy(), len_t, len_l,
                                                           d_cls, d_rec)
    loss_rec, _, _, pred = m(x.cuda(), xu.cuda(), mask_ratio=0.8, noise=(n1, n2), grid=(len_t, len_l), drop_factors=(d_cls, d_rec))
    loss = lam * loss_rec + torch.nn.functional.cross_entropy(pred, y.cuda(), reduction="mean", ignore_index=0)
    loss.backward()
    assert abs(float(loss_rec) - float(o_rec)) < 1e-4 * float(o_rec), (float(loss_rec), float(o_rec))
    assert rms_rel(pred.detach().cpu(), o_pred) < 1e-2
    worst, wname = grad_report(m, o_grads)
    print(f"[finetune 144/72] loss_rec {float(loss_rec):.6f} oracle {float(o_rec):.6f}  worst grad rms-rel {worst:.2e} ({wname})")
    assert worst < 3e-2, (worst, wname)


def test_dual_branch_finetuning_loop_learns_separable_classes(tmp_path):
    """The reference's fine-tuning entry point (Model_Finetuning.py:66-240) end to end on synthetic cubes whose class
    is a spectral offset: train loss falls, validation OA ends far above chance, the checkpoint has DualViT's keys."""
    from hsimae_amd import dual_branch_finetuning
    rng = np.random.default_rng(0)
    n_lab, n_unl, bands, classes = 96, 160, 32, 3
    gt = np.tile(np.arange(1, classes + 1), n_lab // classes)
    ramp = np.linspace(0, 1, bands, dtype=np.float32)

    def cube(c):
        base = 0.25 + 0.2 * c * ramp if c % 2 else 0.75 - 0.2 * c * ramp
        return np.clip(base[None, None, :] + 0.05 * rng.standard_normal((9, 9, bands)).astype(np.float32), 0, 1)

    data_list = [cube(int(c)) for c in gt]
    unlabeled = [cube(int(rng.integers(1, classes + 1))) for _ in range(n_unl)]
    with contextlib.redirect_stdout(io.StringIO()):
        val_value, tr_loss, va_loss = dual_branch_finetuning(
            data_list, list(range(n_lab)), unlabeled, gt, str(tmp_path), "ft.pkl", lr=2e-3, wd=5e-3, depth=4, dim=64,
            dec_depth=1, dec_dim=32, s_depth=2, epochs=8, mask_ratio=0.5, lamda=5, batch_size=16, log=lambda *_: None)
    print(f"[finetune loop] train loss {tr_loss[0]:.3f} -> {tr_loss[-1]:.3f}, val loss {va_loss[0]:.3f} -> {va_loss[-1]:.3f}, "
          f"OA/AA/kappa {val_value[0]:.3f}/{val_value[1]:.3f}/{val_value[2]:.3f}")
    assert tr_loss[-1] < tr_loss[0] and va_loss[-1] < va_loss[0]
    assert val_value[0] > 0.8
    sd = torch.load(os.path.join(str(tmp_path), "ft.pkl"), map_location="cpu")
    assert "cls_head.weight" in sd and "decoder_pred.bias" in sd and "blocks_1.0.attn.q.weight" in sd
    # Model_Finetuning.test_model on the saved checkpoint: HSIViT over fresh cubes of the same classes
    from hsimae_amd import test_model
    test_gt = np.tile(np.arange(1, classes + 1), 20).reshape(6, 10)
    test_cubes = [cube(int(c)) for c in test_gt.reshape(-1)]
    with contextlib.redirect_stdout(io.StringIO()):
        oa, aa, kappa, ca, pred_map = test_model(test_cubes, test_gt, test_gt, str(tmp_path), "ft.pkl", depth=4, dim=64, s_depth=2)
    print(f"[test_model] OA/AA/kappa {oa:.3f}/{aa:.3f}/{kappa:.3f}")
    assert pred_map.shape == test_gt.shape and oa > 0.8 and len(ca) == classes


def test_hsivit_evaluates_a_dualvit_checkpoint():
    """Model_Finetuning.test_model's flow (:243-300): the fine-tuned DualViT state_dict is loaded key-filtered into HSIViT
    (encoder + head, 385 keys), whose logits must be DualViT's own eval logits — and the oracle's."""
    from hsimae_amd import DualViT, HSIViT
    cfg = O.OracleConfig(bands=96)
    state = O.init_state(cfg, seed=4, std=0.02)
    g = torch.Generator().manual_seed(8)
    state["cls_head.weight"] = torch.randn(16, 128 * 12, generator=g) * 0.02
    state["cls_head.bias"] = torch.randn(16, generator=g) * 0.05
    kw = dict(img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, num_class=16, embed_dim=128, depth=12, num_heads=8,
              s_depth=9, trunc_init=True)
    d = quiet(DualViT, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, **kw)
    d.load_state_dict(state)
    v = quiet(HSIViT, **kw)
    model_dict = v.state_dict()
    model_dict.update({k: t for k, t in d.state_dict().items() if k in model_dict})
    assert len(model_dict) == 385
    v.load_state_dict(model_dict)
    d, v = d.cuda().eval(), v.cuda().eval()
    x = torch.rand(24, 1, 96, 9, 9, generator=g)
    ref_pred, _ = O.dualvit_classify(state, cfg, x)
    pd, pv = d(x.cuda()).cpu(), v(x.cuda()).cpu()
    assert torch.equal(pd, pv)
    assert rms_rel(pv, ref_pred) < 1e-2
    v.train()
    with pytest.raises(NotImplementedError):
        v(x.cuda())
