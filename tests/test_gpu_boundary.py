"""Boundary behaviour of hsimae_amd.HSIMAE on a real MI355X:

  * the config-1 pins recorded from the reference are consumed directly by the HIP path (c1_summary, c1_refscale,
    c1_trajectory): loss, 532 gradient L2 norms, stage checksums, the 10-step AdamW trajectory;
  * autograd patterns the reference's nn.Module allows: two forwards before one backward, a monitoring forward between
    forward and backward, zero_grad(set_to_none=False), a foreign `.grad`, a second backward (raises, never silent);
  * forward_encoder / forward_decoder / forward_loss compose under autograd like the reference's (Models.py:975-993);
  * widths the reference's own function defaults use (64/48, 144/72): construct, run and match the oracle;
  * a model on a non-default device (needs 2 GPUs; skipped on the 1-GPU box).
"""
import contextlib
import io
import json
import os
import random

import numpy as np
import pytest
import torch

from hsimae_amd import HSIMAE
from oracle import hsimae_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def base48(seed, dev=DEV, **kw):
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=48, b_patch_size=8, embed_dim=128, depth=12, num_heads=8,
                   s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True, **kw)
    return m.to(dev)


def rms_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def perturb_like_fixture(m, seed, std=0.2):
    """tests/golden/make_golden.py::perturb (our generator walk, values only)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n in ("pos_embed", "decoder_pos_embed", "mask_token"):
                continue
            if "norm" in n and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif n == "patch_embed.proj.weight":
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(std * torch.randn(p.shape, generator=g) / (p.shape[1] ** 0.5) * 4)


# --------------------------------------------------------------------------- reference pins at config 1
def _check_against_summary(s, z, m, x, loss_gate, norm_gate):
    n1, n2 = torch.from_numpy(z["noise_1"]), torch.from_numpy(z["noise_2"])
    loss, pred, mask = m(x.to(DEV), 0.75, noise=(n1, n2), grid=(s["len_t"], s["len_l"]))
    loss.backward()
    rel = abs(loss.item() - s["loss_fp32"]) / s["loss_fp32"]
    assert rel <= loss_gate, f"loss {loss.item()} vs reference {s['loss_fp32']} (rel {rel:.2e})"
    # recons images: mask image exact (sum), prediction image checksums (sum|.|, L2) of the reference
    assert float(mask.sum().item()) == s["mask_img_sum"]
    pi = pred.double()
    assert abs(float(pi.abs().sum()) - s["pred_img"][1]) <= 5e-3 * s["pred_img"][1]
    assert abs(float(pi.pow(2).sum().sqrt()) - s["pred_img"][2]) <= 5e-3 * s["pred_img"][2]
    named = dict(m.named_parameters())
    worst = ("", 0.0)
    for k, ref in s["grad_l2"].items():
        got = float(named[k].grad.double().norm())
        if k.endswith("attn.k.bias"):            # true gradient is exactly zero: the reference holds rounding noise only
            assert got <= 5e-2 * s["grad_l2"][k.replace(".k.bias", ".q.bias")] + 1e-7, k
            continue
        e = abs(got - ref) / max(ref, 1e-12)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] <= norm_gate, f"gradient L2 norm of {worst[0]} off by {worst[1]:.2e}"
    return rel, worst


def test_c1_summary_consumed_directly_perturbed_weights():
    """F5 (tests/golden/c1_summary.*, N = 16, weights ~3.5x the reference's scale: the stress case).  Loss gate 1e-3:
    rounding the weights to bf16 alone moves this loss by +-1..4e-4 (Numerics)."""
    s = json.load(open(os.path.join(G, "c1_summary.json")))
    z = np.load(os.path.join(G, "c1_summary.npz"))
    torch.manual_seed(0); random.seed(0)
    m = base48(0)
    perturb_like_fixture(m, s["perturb_seed"])
    torch.manual_seed(s["x_seed"])
    x = torch.rand(s["N"], 1, 48, 9, 9)
    _check_against_summary(s, z, m, x, 1e-3, 3e-2)
    # stage samples of the reference: latent rows of the first 4 cubes, pred of the first 2
    lat, _, ids_r, ids_k = m.forward_encoder(x.to(DEV), 0.75, noise=(torch.from_numpy(z["noise_1"]), torch.from_numpy(z["noise_2"])),
                                             grid=(2, 7))
    assert torch.equal(ids_k.cpu()[:, :], torch.from_numpy(z["ids_keep"].astype(np.int64)))
    assert rms_rel(lat.detach()[:4], torch.from_numpy(z["latent"])) <= 5e-3


def test_loss_error_is_the_bf16_operand_rounding():
    """VERDICT r03 weak spot 1: with non-degenerate weights (the F5 stress fixture, ~3.5x the reference's scale) the HIP loss is
    only within ~1e-4 of the fp32 reference.  This test shows that gap IS the rounding of the matrix-product operands to bf16 and
    nothing else: the oracle re-run with its operands rounded where the kernels round theirs (`oracle.operands_bf16`: LayerNorm
    outputs, q | k | v, softmax numerators, attention output, gate product, patch values, weights; fp32 everywhere else) lands
    on the HIP loss an order of magnitude closer than the fp32 oracle does.  Gates: |HIP - bf16-operand oracle| <= 3e-5 relative
    (accumulation order, v_exp / v_rcp / v_rsq at 1 ulp: what is left), <= 1e-3 to the fp32 reference record, and the rounding
    itself is visible: |bf16-operand oracle - fp32 oracle| >= 3e-5."""
    s = json.load(open(os.path.join(G, "c1_summary.json")))
    z = np.load(os.path.join(G, "c1_summary.npz"))
    torch.manual_seed(0); random.seed(0)
    m = base48(0)
    perturb_like_fixture(m, s["perturb_seed"])
    torch.manual_seed(s["x_seed"])
    x = torch.rand(s["N"], 1, 48, 9, 9)
    P = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    cfg = O.OracleConfig(bands=48)
    args = (P, cfg, x, z["noise_1"], z["noise_2"], s["len_t"], s["len_l"])
    l32 = O.forward(*args)[0].item()
    with O.operands_bf16():
        lb = O.forward(*args)[0].item()
    assert abs(l32 - s["loss_fp32"]) <= 2e-6 * s["loss_fp32"]                 # the oracle is the reference here
    with torch.no_grad():
        loss, _, _ = m(x.to(DEV), 0.75, noise=(torch.from_numpy(z["noise_1"]), torch.from_numpy(z["noise_2"])),
                       grid=(s["len_t"], s["len_l"]))
    lh = loss.item()
    e_b, e_32, gap = abs(lh - lb) / lb, abs(lh - l32) / l32, abs(lb - l32) / l32
    print(f"[bf16-operand oracle] HIP {lh:.7f}  bf16-operand oracle {lb:.7f} (rel {e_b:.2e})  fp32 oracle {l32:.7f} (rel {e_32:.2e}); "
          f"oracle-to-oracle {gap:.2e}")
    assert gap >= 3e-5
    assert e_32 <= 1e-3
    assert e_b <= 3e-5 and e_b <= 0.5 * gap


def test_gradient_error_is_mostly_the_forward_operand_rounding():
    """VERDICT r03 weak spot 2 (gradient gates of 2e-2 against a suggested 3e-3): at config 1 the gradients of the oracle run with
    bf16-rounded FORWARD operands and an fp32 backward (`oracle.operands_bf16`) are as far from the fp32 reference as the HIP
    gradients are (profiles/r04_grad_error_c1.txt: median 3.7e-3, worst 9.4e-3 over 503 tensors for both) — the error is the
    rounding of the forward operands propagated through the backward, not the backward kernels.  What the backward kernels add on
    top (their own bf16 operands: dO, dq|dk|dv, dh1|dh3, g) is HIP vs that oracle, gated here per family: LayerNorm parameters and
    biases <= 2e-3, weight matrices <= 3e-3 (one bf16 rounding of each weight-gradient operand: 2^-9), q / k projections
    <= 2e-2 (P, dS, dq and dk are bf16 in the attention backward, and these gradients are small differences of large terms;
    measured 1.2e-2 at this N = 16, 7.2e-3 at N = 64)."""
    cfg = O.OracleConfig(bands=48)
    state = O.init_state(cfg, seed=0, std=0.02)
    N = 16
    g = torch.Generator().manual_seed(0)
    x = torch.rand(N, 1, 48, 9, 9, generator=g)
    n1, n2 = torch.rand(N, cfg.T, generator=g), torch.rand(N, 9, generator=g)
    args = (state, cfg, x, n1.numpy(), n2.numpy(), 2, 7)
    _, _, _, g32 = O.forward_backward(*args)
    with O.operands_bf16():
        lb, _, _, gb = O.forward_backward(*args)
    m = base48(0)
    m.load_state_dict(state)
    loss, _, _ = m(x.to(DEV), 0.75, noise=(n1, n2), grid=(2, 7))
    loss.backward()
    assert abs(loss.item() - lb.item()) <= 1e-5 * lb.item()
    named = dict(m.named_parameters())
    worst = {"vec": ("", 0.0), "mat": ("", 0.0), "qk": ("", 0.0)}
    fwd_share = []
    for k, ref in gb.items():
        if k.endswith("attn.k.bias"):
            continue
        e = rms_rel(named[k].grad, ref)
        kind = "qk" if (".attn.q." in k or ".attn.k." in k) else ("mat" if ref.dim() >= 2 else "vec")
        if e > worst[kind][1]:
            worst[kind] = (k, e)
        fwd_share.append((rms_rel(ref, g32[k]), rms_rel(named[k].grad, g32[k])))
    print(f"[grad error vs bf16-operand oracle] worst vector {worst['vec']}, matrix {worst['mat']}, q/k {worst['qk']}")
    assert worst["vec"][1] <= 2e-3 and worst["mat"][1] <= 3e-3 and worst["qk"][1] <= 2e-2, worst
    # the forward rounding alone explains the distance to fp32: medians within 25 % of each other
    a = sorted(v[0] for v in fwd_share)[len(fwd_share) // 2]
    b = sorted(v[1] for v in fwd_share)[len(fwd_share) // 2]
    assert 0.75 * b <= a <= 1.25 * b, (a, b)


def test_c1_config1_reference_scale_n64():
    """BASELINE.json configs[0] exactly, the reference's weights after construction (seed 0): loss <= 1e-4 relative
    (north_star), every gradient L2 norm <= 2e-2, four full gradient tensors RMS-relative <= 2e-2."""
    s = json.load(open(os.path.join(G, "c1_refscale.json")))
    z = np.load(os.path.join(G, "c1_refscale.npz"))
    m = base48(s["model_seed"])
    torch.manual_seed(s["x_seed"])
    x = torch.rand(s["N"], 1, 48, 9, 9)
    rel, worst = _check_against_summary(s, z, m, x, 1e-4, 2e-2)
    named = dict(m.named_parameters())
    for key, name in (("g_blocks0_w2", "blocks.0.mlp.w2.weight"), ("g_b1_0_q", "blocks_1.0.attn.q.weight"),
                      ("g_dec7_w1", "decoder_blocks.7.mlp.w1.weight"), ("g_pe", "patch_embed.proj.weight")):
        e = rms_rel(named[name].grad, torch.from_numpy(z[key]))
        assert e <= 2e-2, (name, e)
    print(f"config 1: loss rel {rel:.2e}; worst grad norm {worst}")


@pytest.mark.parametrize("tag,precision,loss_gate,norm_gate,rms_gate",
                         [("c3", "bf16", 1e-4, 2e-2, 2e-2), ("c5", "bf16", 1e-4, 2e-2, 2e-2), ("c5", "fp8", 1e-3, 6e-2, 1.5e-1),
                          ("c3", "fp8", 1e-4, 2e-2, 2e-2)])      # Large under "fp8": below embed_dim 512 the bf16 kernels run (same gates as bf16)
def test_wide_configs_match_the_reference_record(tag, precision, loss_gate, norm_gate, rms_gate):
    """HSIMAE-Large (D = 256, 16 heads) and the D = 512 / 32-head / 192-band model against the reference's own record
    (tests/golden/make_golden_wide.py; reference weights after construction).  bf16: the north-star gates (loss 1e-4 relative,
    gradient norms 2e-2).  fp8 (MX e4m3 encoder linears): loss 1e-3, gradient norms 6 %, gradient elements 15 % RMS — the fp8 tolerance of DESIGN.md 4
    (measured: loss 2.4e-5, worst norm 2.8 %)."""
    s = json.load(open(os.path.join(G, f"{tag}_refscale.json")))
    z = np.load(os.path.join(G, f"{tag}_refscale.npz"))
    torch.manual_seed(s["model_seed"])
    with contextlib.redirect_stdout(io.StringIO()):
        m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=s["bands"], b_patch_size=8, embed_dim=s["dim"], depth=12,
                   num_heads=s["heads"], s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8,
                   norm_pix_loss=True, trunc_init=True).to(DEV)
    m.set_precision(precision)
    torch.manual_seed(s["x_seed"])
    x = torch.rand(s["N"], 1, s["bands"], 9, 9)
    n1, n2 = torch.from_numpy(z["noise_1"]), torch.from_numpy(z["noise_2"])
    loss, pred, mask = m(x.to(DEV), 0.75, noise=(n1, n2), grid=(s["len_t"], s["len_l"]))
    loss.backward()
    rel = abs(loss.item() - s["loss_fp32"]) / s["loss_fp32"]
    assert rel <= loss_gate, f"loss {loss.item()} vs reference {s['loss_fp32']} (rel {rel:.2e})"
    assert float(mask.sum().item()) == s["mask_img_sum"]                      # the masking is index work: exact in every precision
    _, _, _, ids_k = m.forward_encoder(x.to(DEV), 0.75, noise=(n1, n2), grid=(s["len_t"], s["len_l"]))
    assert torch.equal(ids_k.cpu(), torch.from_numpy(z["ids_keep"].astype(np.int64)))
    named = dict(m.named_parameters())
    worst = ("", 0.0)
    for k, ref in s["grad_l2"].items():
        got = float(named[k].grad.double().norm())
        if k.endswith("attn.k.bias"):            # true gradient is exactly zero
            assert got <= 5e-2 * s["grad_l2"][k.replace(".k.bias", ".q.bias")] + 1e-7, k
            continue
        e = abs(got - ref) / max(ref, 1e-12)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] <= norm_gate, f"gradient L2 norm of {worst[0]} off by {worst[1]:.2e}"
    for key, name in (("g_blocks0_w2", "blocks.0.mlp.w2.weight"), ("g_b1_0_q", "blocks_1.0.attn.q.weight"),
                      ("g_dec7_w1", "decoder_blocks.7.mlp.w1.weight"), ("g_pe", "patch_embed.proj.weight")):
        ref = torch.from_numpy(z[key])
        e = rms_rel(named[name].grad[:ref.shape[0]], ref)
        assert e <= rms_gate, (name, e)
    print(f"{tag} {precision}: loss rel {rel:.2e}; worst grad norm {worst}")


def test_c1_config1_trajectory_matches_reference():
    """F6 at config 1: 10 AdamW steps (stock torch.optim.AdamW on the same Parameters), each loss <= 1e-3 relative."""
    meta = json.load(open(os.path.join(G, "c1_trajectory.json")))
    z = np.load(os.path.join(G, "c1_trajectory.npz"))
    m = base48(meta["model_seed"])
    torch.manual_seed(meta["x_seed"])
    x = torch.rand(meta["N"], 1, 48, 9, 9).to(DEV)
    nd = ["bias", "norm"]
    groups = [{"params": [p for n, p in m.named_parameters() if not any(k in n for k in nd)], "weight_decay": meta["wd"]},
              {"params": [p for n, p in m.named_parameters() if any(k in n for k in nd)], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=meta["lr"], weight_decay=meta["wd"], betas=tuple(meta["betas"]))
    worst = 0.0
    for i, ref in enumerate(meta["losses"]):
        loss, _, _ = m(x, meta["ratio"], noise=(torch.from_numpy(z[f"n1_{i}"]), torch.from_numpy(z[f"n2_{i}"])),
                       grid=tuple(meta["grids"][i]))
        opt.zero_grad(); loss.backward(); opt.step()
        worst = max(worst, abs(loss.item() - ref) / abs(ref))
        assert abs(loss.item() - ref) <= 1e-3 * abs(ref), (i, loss.item(), ref)
    print(f"trajectory worst rel {worst:.2e}")


# --------------------------------------------------------------------------- autograd patterns
def _inputs(N=6, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, 1, 48, 9, 9, generator=g).to(DEV)
    return x, (torch.rand(N, 6, generator=g), torch.rand(N, 9, generator=g))


def _grads(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def test_two_forwards_then_one_backward_equals_sum_of_separate_backwards():
    m = base48(1)
    perturb_like_fixture(m, 9, std=0.05)
    (x1, nz1), (x2, nz2) = _inputs(6, 5), _inputs(6, 6)
    sep = []
    for x, nz in ((x1, nz1), (x2, nz2)):
        m.zero_grad(set_to_none=True)
        m(x, 0.75, noise=nz, grid=(2, 7))[0].backward()
        sep.append(_grads(m))
    m.zero_grad(set_to_none=True)
    l1 = m(x1, 0.75, noise=nz1, grid=(2, 7))[0]
    with torch.no_grad():                                   # a monitoring forward in between must not disturb anything
        m(x2, 0.75, noise=nz2, grid=(2, 7))
    l2 = m(x2, 0.75, noise=nz2, grid=(2, 7))[0]
    m.eval(); m(x1[:2], 0.75); m.train()
    (l1 + l2).backward()
    both = _grads(m)
    assert len(both) == 532
    for k in both:
        ref = sep[0][k] + sep[1][k]
        # weight gradients are committed with fp32 atomics: equality up to summation order
        assert float((both[k] - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-9, k


def test_second_backward_of_the_same_forward_raises():
    m = base48(1)
    x, nz = _inputs()
    loss = m(x, 0.75, noise=nz, grid=(2, 7))[0]
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already been back-propagated"):
        loss.backward()


def test_grad_accumulation_zero_grad_in_place_and_foreign_grad_tensors():
    m = base48(2)
    x, nz = _inputs()
    m.zero_grad(set_to_none=True)
    m(x, 0.75, noise=nz, grid=(2, 7))[0].backward()
    g1 = _grads(m)
    # accumulate onto existing grads
    m(x, 0.75, noise=nz, grid=(2, 7))[0].backward()
    for k, g in _grads(m).items():
        assert float((g - 2 * g1[k]).abs().max()) <= 2e-4 * float(g1[k].abs().max()) + 1e-9, k
    # zero_grad(set_to_none=False): grads stay the flat-buffer views, zeroed in place
    m.zero_grad(set_to_none=False)
    m(x, 0.75, noise=nz, grid=(2, 7))[0].backward()
    for k, g in _grads(m).items():
        assert float((g - g1[k]).abs().max()) <= 2e-4 * float(g1[k].abs().max()) + 1e-9, k
    # a caller replaces one .grad by its own tensor and clears another: per-parameter assign / accumulate
    named = dict(m.named_parameters())
    own = torch.full_like(named["norm.weight"], 3.0)
    named["norm.weight"].grad = own
    named["decoder_pred.bias"].grad = None
    before = named["blocks.0.mlp.w1.weight"].grad.clone()
    m(x, 0.75, noise=nz, grid=(2, 7))[0].backward()
    assert named["norm.weight"].grad is own
    assert torch.allclose(own, 3.0 + g1["norm.weight"], rtol=1e-3, atol=1e-7)
    assert torch.allclose(named["decoder_pred.bias"].grad, g1["decoder_pred.bias"], rtol=1e-3, atol=1e-7)
    assert torch.allclose(named["blocks.0.mlp.w1.weight"].grad, before + g1["blocks.0.mlp.w1.weight"], rtol=1e-3, atol=1e-8)


def test_sub_entry_points_compose_under_autograd_like_forward():
    """loss = forward_loss(imgs, forward_decoder(forward_encoder(imgs))) gives forward()'s loss and gradients."""
    m = base48(3)
    perturb_like_fixture(m, 4, std=0.05)
    x, nz = _inputs(8, 11)
    m.zero_grad(set_to_none=True)
    loss_ref = m(x, 0.75, noise=nz, grid=(2, 7))[0]
    loss_ref.backward()
    ref = _grads(m)
    m.zero_grad(set_to_none=True)
    latent, mask, ids_restore, ids_keep = m.forward_encoder(x, 0.75, noise=nz, grid=(2, 7))
    assert latent.requires_grad and not mask.requires_grad
    pred = m.forward_decoder(latent, ids_restore)
    loss = m.forward_loss(x, pred, mask)
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    (2.0 * loss).backward()                                  # an upstream factor goes through all three nodes
    got = _grads(m)
    assert set(got) == set(ref)
    worst = max((rms_rel(got[k], 2.0 * ref[k]), k) for k in ref if not k.endswith("attn.k.bias"))
    assert worst[0] <= 2e-2, worst
    # decoder alone from a detached latent: only the decoder's parameters receive gradients, the latent gets one too
    m.zero_grad(set_to_none=True)
    lat2 = latent.detach().clone().requires_grad_(True)
    m.forward_loss(x, m.forward_decoder(lat2, ids_restore), mask).backward()
    named = dict(m.named_parameters())
    assert named["blocks.0.mlp.w1.weight"].grad is None and named["decoder_blocks.0.mlp.w1.weight"].grad is not None
    assert lat2.grad is not None and float(lat2.grad.abs().sum()) > 0


def test_fused_adamw_skips_parameters_without_grad_after_a_partial_backward():
    """FusedAdamW steps the flat gradient buffer; a decoder-only backward after zero_grad() leaves the encoder's
    `.grad` None and the previous step's encoder gradients in the flat buffer.  Those parameters must be skipped exactly as
    torch.optim.AdamW skips them (no update, no weight decay), and a caller-owned `.grad` tensor must be what is applied."""
    from hsimae_amd import FusedAdamW
    m = base48(3)
    opt = FusedAdamW(m, lr=1e-2, weight_decay=5e-2, betas=(0.9, 0.95))
    g = torch.Generator().manual_seed(9)
    x = torch.rand(8, 1, 48, 9, 9, generator=g).to(DEV)
    noise = (torch.rand(8, 6, generator=g), torch.rand(8, 9, generator=g))
    loss, _, _ = m(x, 0.75, noise=noise, grid=(2, 7))
    loss.backward()
    opt.step()                                            # every trainable parameter has a gradient: the usual step
    opt.zero_grad()
    enc_before = {k: p.detach().clone() for k, p in m.named_parameters() if k.startswith(("blocks", "patch_embed", "norm."))}
    dec_before = {k: p.detach().clone() for k, p in m.named_parameters() if k.startswith("decoder_blocks")}
    with torch.no_grad():
        latent, mask, ids_restore, _ = m.forward_encoder(x, 0.75, noise=noise, grid=(2, 7))
    loss = m.forward_loss(x, m.forward_decoder(latent, ids_restore), mask)
    loss.backward()                                       # decoder-only backward: writes the decoder's flat range only
    named = dict(m.named_parameters())
    assert named["blocks.0.mlp.w1.weight"].grad is None and named["decoder_blocks.0.mlp.w1.weight"].grad is not None
    opt.step()
    torch.cuda.synchronize()
    for k, v in enc_before.items():
        assert torch.equal(named[k].detach(), v), f"{k} moved although it had no gradient"
    moved = sum(int(not torch.equal(named[k].detach(), v)) for k, v in dec_before.items())
    assert moved == len(dec_before)
    # a foreign `.grad` (assigned by the caller) is what the step applies: +1 everywhere on one bias -> it moves by ~ -lr
    opt.zero_grad()
    b = named["decoder_blocks.0.mlp.w2.bias"]
    b0 = b.detach().clone()
    b.grad = torch.ones_like(b)
    opt2 = FusedAdamW(m, lr=1e-2, weight_decay=0.0)       # fresh moments: the first Adam step is -lr * sign(g)
    opt2.step()
    torch.cuda.synchronize()
    assert torch.allclose(b.detach(), b0 - 1e-2, atol=1e-5)


def test_fused_adamw_common_path_is_one_launch_and_no_torch_ops():
    """after a whole backward every trainable `.grad` is at home in the flat buffer — FusedAdamW.step() must
    then be ONE library launch: no ATen op at all on the way (round 3 issued a device fill per parameter when gradients were
    missing, and walked 532 `.grad` objects on every step).  Counted with a TorchDispatchMode (every ATen call, views included);
    the partial-backward pattern is allowed one mask upload the first time and nothing afterwards (cached per pattern)."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from hsimae_amd import FusedAdamW

    class Count(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.ops = []

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            self.ops.append(str(func))
            return func(*args, **(kwargs or {}))

    m = base48(3)
    opt = FusedAdamW(m, lr=1e-3, weight_decay=5e-2, betas=(0.9, 0.95))
    g = torch.Generator().manual_seed(9)
    x = torch.rand(8, 1, 48, 9, 9, generator=g).to(DEV)
    noise = (torch.rand(8, 6, generator=g), torch.rand(8, 9, generator=g))
    opt.zero_grad()
    m(x, 0.75, noise=noise, grid=(2, 7))[0].backward()
    opt.step()                                            # the first step binds the optimizer to the flat buffer (moments, group table)
    for strict in (False, True):
        opt.strict = strict
        for _ in range(2):
            opt.zero_grad()
            m(x, 0.75, noise=noise, grid=(2, 7))[0].backward()
            before = {k: p.detach().clone() for k, p in list(m.named_parameters())[2:6]}
            with Count() as c:
                opt.step()
            torch.cuda.synchronize()
            assert c.ops == [], f"strict={strict}: the common path issued torch ops: {c.ops[:8]}"
            assert all(not torch.equal(p.detach(), before[k]) for k, p in list(m.named_parameters())[2:6] if p.requires_grad)
    opt.strict = False
    # decoder-only backward: the encoder's gradients are missing -> one host-built mask, uploaded once and cached
    counts = []
    for _ in range(3):
        opt.zero_grad()
        with torch.no_grad():
            latent, mask, ids_restore, _ = m.forward_encoder(x, 0.75, noise=noise, grid=(2, 7))
        m.forward_loss(x, m.forward_decoder(latent, ids_restore), mask).backward()
        with Count() as c:
            opt.step()
        counts.append(len(c.ops))
    assert counts[0] <= 4 and counts[1] == 0 and counts[2] == 0, counts


def test_fused_adamw_default_sees_a_middle_gradient_replaced_after_backward():
    """the default FusedAdamW must treat `.grad` as torch.optim.AdamW does — a MIDDLE parameter whose
    `.grad` is set to None after the backward is skipped, one whose `.grad` is replaced by a new tensor is stepped with the new
    values (the 4-probe fast path, `strict=False`, is opt-in because it cannot see either)."""
    from hsimae_amd import FusedAdamW
    m = base48(5)
    opt = FusedAdamW(m, lr=1e-2, weight_decay=0.0)
    assert opt.strict
    g = torch.Generator().manual_seed(11)
    x = torch.rand(8, 1, 48, 9, 9, generator=g).to(DEV)
    noise = (torch.rand(8, 6, generator=g), torch.rand(8, 9, generator=g))
    opt.zero_grad()
    m(x, 0.75, noise=noise, grid=(2, 7))[0].backward()
    named = dict(m.named_parameters())
    skipped, replaced = named["blocks_2.4.mlp.w1.weight"], named["decoder_blocks.3.attn.proj.weight"]
    s0, r0 = skipped.detach().clone(), replaced.detach().clone()
    skipped.grad = None
    replaced.grad = torch.ones_like(replaced)               # first Adam step with fresh moments: -lr * sign(g) = -1e-2 everywhere
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(skipped.detach(), s0)
    assert torch.allclose(replaced.detach(), r0 - 1e-2, atol=1e-5)


@pytest.mark.parametrize("dim,dec_dim,bands", [(64, 48, 32), (144, 72, 32)])
def test_reference_default_widths_construct_and_match_oracle(dim, dec_dim, bands):
    """Model_Pretraining.py:57-58 (dim 64, dec_dim 48) and Model_Finetuning.py:66-67 (144 / 72): widths that are
    multiples of 8 / 16 but not of 32."""
    cfg = O.OracleConfig(bands=bands, embed_dim=dim, depth=12, num_heads=dim // 16, s_depth=6, decoder_embed_dim=dec_dim,
                         decoder_depth=2, decoder_num_heads=dec_dim // 8)
    state = O.init_state(cfg, seed=0, std=0.02)
    N = 10
    g = torch.Generator().manual_seed(3)
    x = torch.rand(N, 1, bands, 9, 9, generator=g)
    n1, n2 = torch.rand(N, cfg.T, generator=g), torch.rand(N, 9, generator=g)
    lt, ll = HSIMAE.grid_candidates(cfg.T, 9, 0.5)[0]
    ref_loss, _, ref_mask, ref_grads = O.forward_backward(state, cfg, x, n1.numpy(), n2.numpy(), lt, ll)
    with contextlib.redirect_stdout(io.StringIO()):
        m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=bands, b_patch_size=8, embed_dim=dim, depth=12,
                   num_heads=dim // 16, s_depth=6, decoder_embed_dim=dec_dim, decoder_depth=2, decoder_num_heads=dec_dim // 8,
                   norm_pix_loss=True, trunc_init=True)
    m.load_state_dict(state)
    m = m.to(DEV)
    loss, pred, mask = m(x.to(DEV), 0.5, noise=(n1, n2), grid=(lt, ll))
    loss.backward()
    assert torch.equal(mask.cpu(), ref_mask)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * ref_loss.item()
    named = dict(m.named_parameters())
    worst = max((rms_rel(named[k].grad, ref_grads[k]), k) for k in ref_grads if not k.endswith("attn.k.bias"))
    assert worst[0] <= 3e-2, worst


@pytest.mark.parametrize("dim,bands,N,prec", [(128, 48, 64, "bf16"), (256, 96, 16, "bf16"), (512, 192, 4, "fp8")])
def test_deterministic_mode_is_bit_reproducible(dim, bands, N, prec):
    """SURVEY 5 (determinism): with `deterministic` on, two runs on the same inputs give bit-identical gradients for all 532
    tensors (fused and layer-at-a-time schedules, two streams); the result agrees with the fp32-atomics path to rounding."""
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(1)
        m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=bands, b_patch_size=8, embed_dim=dim, depth=12, num_heads=dim // 16,
                   s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True)
    m = m.to(DEV).set_precision(prec)
    perturb_like_fixture(m, 3, std=0.05)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(N, 1, bands, 9, 9, generator=g).to(DEV)
    T = bands // 8
    nz = (torch.rand(N, T, generator=g), torch.rand(N, 9, generator=g))
    grid = HSIMAE.grid_candidates(T, 9, 0.75)[0]

    def run():
        m.zero_grad(set_to_none=True)
        loss = m(x, 0.75, noise=nz, grid=grid)[0]
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), _grads(m)

    m.deterministic = True
    l1, g1 = run()
    l2, g2 = run()
    assert l1 == l2 and len(g1) == 532
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    m.deterministic = False
    _, g3 = run()
    worst = max((float((g1[k] - g3[k]).abs().max() / g3[k].abs().max().clamp_min(1e-20)), k) for k in g1 if not k.endswith("attn.k.bias"))
    assert worst[0] < 2e-4, worst


def test_repeated_steps_agree_with_the_deterministic_result():
    """Race hunt at batch 64, where every sample is the first one of its decoder workgroup: 60 steps in fp32-atomics mode, each
    against the deterministic-mode gradients.  (A missing barrier after the LDS staging of LayerNorm-1's gamma / beta in the decoder
    attention backward showed up here as a 1e-3-level deviation of one block's q / k / v gradients in ~3 % of the steps.)"""
    m = base48(1)
    perturb_like_fixture(m, 3, std=0.05)
    x, nz = _inputs(64, seed=5)

    def run():
        m.zero_grad(set_to_none=True)
        m(x, 0.75, noise=nz, grid=(2, 7))[0].backward()
        torch.cuda.synchronize()
        return _grads(m)

    m.deterministic = True
    ref = run()
    m.deterministic = False
    for i in range(60):
        g = run()
        worst = max((float((g[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-20)), k) for k in ref if not k.endswith("attn.k.bias"))
        assert worst[0] < 2e-4, (i, worst)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_model_on_a_non_default_device():
    torch.cuda.set_device(0)
    m0, m1 = base48(7, "cuda:0"), base48(7, "cuda:1")
    g = torch.Generator().manual_seed(5)
    x = torch.rand(6, 1, 48, 9, 9, generator=g)
    nz = (torch.rand(6, 6, generator=g), torch.rand(6, 9, generator=g))
    l0 = m0(x.to("cuda:0"), 0.75, noise=nz, grid=(2, 7))[0]
    l1 = m1(x.to("cuda:1"), 0.75, noise=nz, grid=(2, 7))[0]       # current device is still 0
    l0.backward(); l1.backward()
    assert abs(l0.item() - l1.item()) <= 1e-6 * abs(l0.item())
    a, b = dict(m0.named_parameters()), dict(m1.named_parameters())
    assert all(b[k].grad.device.index == 1 for k in b if b[k].grad is not None)
    assert rms_rel(b["blocks.0.mlp.w2.weight"].grad, a["blocks.0.mlp.w2.weight"].grad) <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["HSIMAE_ATTN_BWD_RECOMPUTE=0", "HSIMAE_FUSED_MLP=0"])
def test_config1_record_under_the_schedule_switches(switch):
    """The config-1 record of the reference (loss, 532 gradient norms) must hold on the fused attention-half backward fed by
    saved q|k|v and on the layer-at-a-time MLP half (each in a child process with the switch set for the whole run)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    k, v = switch.split("=")
    env = dict(os.environ)
    env[k] = v
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_boundary.py", "-m", "gpu", "-x", "-q", "-k",
                        "test_c1_config1_reference_scale_n64 or test_c1_base48"], capture_output=True, text=True, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
