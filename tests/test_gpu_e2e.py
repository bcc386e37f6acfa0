"""End-to-end parity of hsimae_amd.HSIMAE (HIP path) on a real MI355X.

Checked against (a) fixtures recorded from the reference (tiny model: every stage, every gradient; 10-step
AdamW trajectory) and (b) the CPU oracle on seeded inputs at Base size (config C1), plus size-independent
properties at the full benchmark size (C2: N=4096, 96 bands).
Tolerances (stated per assert): masks / indices bit-exact; loss <= 1e-4 relative (BASELINE.json north_star);
activations and gradients: bf16 GEMM operands with fp32 accumulation => RMS-relative <= 1e-2.
"""
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


def build(cfg: O.OracleConfig, state: dict):
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=cfg.bands, b_patch_size=8, embed_dim=cfg.embed_dim,
               depth=cfg.depth, num_heads=cfg.num_heads, s_depth=cfg.s_depth, decoder_embed_dim=cfg.decoder_embed_dim,
               decoder_depth=cfg.decoder_depth, decoder_num_heads=cfg.decoder_num_heads, norm_pix_loss=cfg.norm_pix_loss,
               trunc_init=True)
    m.load_state_dict(state)
    return m.to(DEV)


def rms_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def grad_err(named, ref_grads, k):
    """RMS-relative gradient error.  attn.k.bias has an exactly-zero true gradient (softmax is invariant to a
    per-query constant), so the reference holds only rounding noise there: compare it on the q.bias scale."""
    g = named[k].grad
    assert g is not None, k
    if k.endswith("attn.k.bias"):
        scale = ref_grads[k.replace(".k.bias", ".q.bias")].double().pow(2).mean().sqrt()
        return float((g.double().cpu() - ref_grads[k].double()).pow(2).mean().sqrt() / scale)
    return rms_rel(g, ref_grads[k])


def max_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


TINY = O.OracleConfig(bands=32, embed_dim=32, depth=3, num_heads=2, s_depth=2, decoder_embed_dim=32,
                      decoder_depth=2, decoder_num_heads=4)


@pytest.mark.parametrize("tag", ["r50", "r75"])
def test_tiny_model_against_reference_fixture(tag):
    z = np.load(os.path.join(G, f"tiny_model_{tag}.npz"))
    state = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
    m = build(TINY, state)
    x = torch.from_numpy(z["x"]).to(DEV)
    lt, ll = (int(v) for v in z["len_tl"])
    noise = (torch.from_numpy(z["noise_1"]), torch.from_numpy(z["noise_2"]))
    loss, pred, mask = m(x, 0.5 if tag == "r50" else 0.75, noise=noise, grid=(lt, ll))
    loss.backward()
    torch.cuda.synchronize()
    ref_loss = float(z["loss"])
    print(f"[tiny {tag}] loss {loss.item():.7f} ref {ref_loss:.7f} rel {abs(loss.item() - ref_loss) / ref_loss:.2e}")
    assert torch.equal(mask.cpu(), torch.from_numpy(z["mask_img"]).float())            # bit-exact selection
    assert abs(loss.item() - ref_loss) <= 1e-3 * ref_loss     # tiny widths (K=32) average fewer bf16 roundings than Base
    assert rms_rel(pred, torch.from_numpy(z["pred_img"])) < 2e-2
    worst = ("", 0.0)
    named = dict(m.named_parameters())
    ref_grads = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad_")}
    for k in ref_grads:
        r = grad_err(named, ref_grads, k)
        if r > worst[1]:
            worst = (k, r)
        assert r < 5e-2, (k, r)
    print(f"[tiny {tag}] worst grad rms-rel {worst}")
    for k in ("pos_embed", "decoder_pos_embed", "mask_token"):
        assert dict(m.named_parameters())[k].grad is None


def test_forward_encoder_outputs_and_ids_dtype():
    z = np.load(os.path.join(G, "tiny_model_r50.npz"))
    state = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
    m = build(TINY, state)
    lt, ll = (int(v) for v in z["len_tl"])
    noise = (torch.from_numpy(z["noise_1"]), torch.from_numpy(z["noise_2"]))
    latent, mask, ids_restore, ids_keep = m.forward_encoder(torch.from_numpy(z["x"]).to(DEV), 0.5, noise=noise, grid=(lt, ll))
    assert ids_keep.dtype == torch.int64 and ids_restore.dtype == torch.int64
    assert torch.equal(ids_keep.cpu(), torch.from_numpy(z["tap_ids_keep"].astype(np.int64)))
    assert torch.equal(ids_restore.cpu(), torch.from_numpy(z["tap_ids_restore"].astype(np.int64)))
    assert torch.equal(mask.cpu(), torch.from_numpy(z["tap_mask"]))
    assert rms_rel(latent, torch.from_numpy(z["tap_latent"])) < 1e-2
    assert (m.len_t, m.len_l) == (lt, ll)


@pytest.mark.parametrize("std,loss_gate", [(0.02, 1e-4), (0.08, 1e-3)])
def test_c1_base48_against_oracle_loss_latent_grads(std, loss_gate):
    """Config C1 (Base, 48 bands, N = 64) against the fp32 CPU oracle.

    std = 0.02 is the reference's own weight scale (trunc_normal_(std=.02), Models.py:452): the north_star gate
    (loss within 1e-4 relative) applies there and is met with ~10x margin (scripts/loss_error_sweep.py: |err| <=
    1.6e-5 over 6 seeds).  std = 0.08 is a stress case with 4x larger weights, where rounding the WEIGHTS to
    bf16 alone moves the loss by +-1..4e-4 (sign varies with the seed), so its gate is 1e-3; activations and
    gradients keep the same gates in both cases."""
    cfg = O.OracleConfig(bands=48)
    state = O.init_state(cfg, seed=1, std=std)
    N, lt, ll = 64, 2, 7
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(N, 1, 48, 9, 9, generator=g)
    n1, n2 = torch.rand(N, 6, generator=g), torch.rand(N, 9, generator=g)
    taps = {}
    ref_loss, ref_pred, ref_mask, ref_grads = O.forward_backward(state, cfg, x, n1.numpy(), n2.numpy(), lt, ll, taps)
    m = build(cfg, state)
    loss, pred, mask = m(x.to(DEV), 0.75, noise=(n1, n2), grid=(lt, ll))
    loss.backward()
    torch.cuda.synchronize()
    rel = abs(loss.item() - ref_loss.item()) / ref_loss.item()
    print(f"[C1 std={std}] loss {loss.item():.7f} oracle {ref_loss.item():.7f} rel {rel:.2e}")
    assert torch.equal(mask.cpu(), ref_mask)
    assert rel <= loss_gate
    assert rms_rel(pred, ref_pred) < 1e-2
    lat, _, _, keep = m.forward_encoder(x.to(DEV), 0.75, noise=(n1, n2), grid=(lt, ll))
    assert torch.equal(keep.cpu(), taps["ids_keep"])
    r_lat, m_lat = rms_rel(lat, taps["latent"].detach()), max_rel(lat, taps["latent"].detach())
    print(f"[C1 std={std}] latent rms-rel {r_lat:.2e} max-rel {m_lat:.2e}")
    assert r_lat < 5e-3 and m_lat < 3e-2
    named = dict(m.named_parameters())
    worst = ("", 0.0)
    for k, gref in ref_grads.items():
        r = grad_err(named, ref_grads, k)
        if r > worst[1]:
            worst = (k, r)
        assert r < 2e-2, (k, r)
    print(f"[C1 std={std}] worst grad rms-rel {worst}")


def test_band_fastest_strided_input_matches_contiguous():
    cfg = O.OracleConfig(bands=48)
    m = build(cfg, O.init_state(cfg, seed=2, std=0.05))
    g = torch.Generator().manual_seed(5)
    x = torch.rand(8, 1, 48, 9, 9, generator=g).to(DEV)
    n = (torch.rand(8, 6, generator=g), torch.rand(8, 9, generator=g))
    xs = x[:, 0].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).unsqueeze(1)     # HSIdataset4PT strides
    assert not xs.is_contiguous()
    with torch.no_grad():
        a = m(x, 0.75, noise=n, grid=(2, 7))
        b = m(xs, 0.75, noise=n, grid=(2, 7))
    assert a[0].item() == b[0].item() and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


def test_gradient_linearity_accumulation_and_zero_grad():
    cfg = O.OracleConfig(bands=48)
    m = build(cfg, O.init_state(cfg, seed=3, std=0.05))
    g = torch.Generator().manual_seed(6)
    x = torch.rand(16, 1, 48, 9, 9, generator=g).to(DEV)
    n = (torch.rand(16, 6, generator=g), torch.rand(16, 9, generator=g))
    p = m.decoder_blocks[3].mlp.w1.weight
    m(x, 0.75, noise=n, grid=(2, 7))[0].backward()
    g1 = p.grad.clone()
    m.zero_grad()
    assert p.grad is None
    (3.0 * m(x, 0.75, noise=n, grid=(2, 7))[0]).backward()
    assert rms_rel(p.grad, 3 * g1) < 1e-4                           # chain rule through d(loss)
    m(x, 0.75, noise=n, grid=(2, 7))[0].backward()                  # no zero_grad: accumulates like autograd
    assert rms_rel(p.grad, 4 * g1) < 1e-4


def test_training_trajectory_matches_reference_fixture():
    z = np.load(os.path.join(G, "trajectory.npz"))
    meta = json.load(open(os.path.join(G, "trajectory.json")))
    cfg = O.OracleConfig(**meta["cfg"])
    state = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
    m = build(cfg, state)
    nd = ["bias", "norm"]
    groups = [{"params": [p for n, p in m.named_parameters() if not any(k in n for k in nd)], "weight_decay": meta["wd"]},
              {"params": [p for n, p in m.named_parameters() if any(k in n for k in nd)], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=meta["lr"], weight_decay=meta["wd"], betas=tuple(meta["betas"]))   # Model_Pretraining.py:80-86
    x = torch.from_numpy(z["x"]).to(DEV)
    losses = []
    for i, ref in enumerate(meta["losses"]):
        noise = (torch.from_numpy(z[f"n1_{i}"]), torch.from_numpy(z[f"n2_{i}"]))
        loss, _, _ = m(x, meta["ratio"], noise=noise, grid=tuple(meta["grids"][i]))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    print("[traj] hip", [f"{v:.5f}" for v in losses])
    print("[traj] ref", [f"{v:.5f}" for v in meta["losses"]])
    for got, ref in zip(losses, meta["losses"]):
        assert abs(got - ref) <= 5e-3 * ref         # 10 optimiser steps amplify bf16 rounding; fp32 oracle pins 1e-5
    # the checkpoint is the reference's wire format
    sd = m.state_dict()
    assert list(sd.keys()) == list(state.keys()) and all(v.dtype == torch.float32 for v in sd.values())


def test_rng_streams_python_random_then_device_rand():
    cfg = O.OracleConfig(bands=96)
    m = build(cfg, O.init_state(cfg, seed=4))
    x = torch.rand(4, 1, 96, 9, 9, device=DEV)
    random.seed(0)
    torch.manual_seed(0)
    seq = []
    for _ in range(5):
        with torch.no_grad():
            m(x, 0.75)
        seq.append([m.len_t, m.len_l])
    meta = json.load(open(os.path.join(G, "masking.json")))
    assert seq == meta["draws"]["0"]                # same python-random consumption as the reference
    torch.manual_seed(0)
    with torch.no_grad():
        a = m(x, 0.75, grid=(3, 9))[2]
    torch.manual_seed(0)
    n1, n2 = torch.rand(4, 12, device=DEV), torch.rand(4, 9, device=DEV)
    with torch.no_grad():
        b = m(x, 0.75, noise=(n1, n2), grid=(3, 9))[2]
    assert torch.equal(a, b)                         # noise_1 then noise_2 from the device generator


def test_c2_full_size_properties_n4096():
    cfg = O.OracleConfig(bands=96)
    m = build(cfg, O.init_state(cfg, seed=5, std=0.05))
    N = 4096
    torch.manual_seed(7)
    x = torch.rand(N, 1, 96, 9, 9, device=DEV)
    n1, n2 = torch.rand(N, 12), torch.rand(N, 9)
    for grid in ((3, 9), (9, 3)):
        m.zero_grad()
        loss, pred, mask = m(x, 0.75, noise=(n1, n2), grid=grid)
        loss.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(loss) and torch.isfinite(pred).all()
        assert float(mask.sum()) == N * (108 - 27) * 72
        k2, r2, m2 = O.mask_from_noise(n1.numpy(), n2.numpy(), *grid)
        assert torch.equal(mask.cpu(), O.unpatchify(torch.from_numpy(m2).unsqueeze(2).repeat(1, 1, 72), cfg))
        for name, p in m.named_parameters():
            if p.requires_grad and name != "mask_token":
                assert p.grad is not None and torch.isfinite(p.grad).all(), name
    # batch-size independence: the first 64 samples alone give the same per-sample predictions
    with torch.no_grad():
        small = m(x[:64], 0.75, noise=(n1[:64], n2[:64]), grid=(9, 3))[1]
    assert rms_rel(small, pred[:64]) < 1e-6


# 64 / 80 bands: 72 / 90 decoder tokens on the 7-tile kernels — key tiles that are partly or wholly padding, fewer query tiles
@pytest.mark.parametrize("bands,grid", [(48, (2, 7)), (96, (3, 9)), (192, (6, 9)), (64, (3, 6)), (80, (3, 7))])
def test_fused_decoder_matches_layerwise_decoder(bands, grid):
    """The fused decoder-block kernels (fused_dec.hip) against the layer-at-a-time kernels, same inputs:
    forward loss / predictions and every gradient.  Both compute in bf16-operand / fp32-accumulate arithmetic;
    they differ only in where intermediates are rounded, so the gate is far tighter than the oracle gate."""
    cfg = O.OracleConfig(bands=bands)
    m = build(cfg, O.init_state(cfg, seed=11, std=0.06))
    N = 24
    g = torch.Generator().manual_seed(21)
    x = torch.rand(N, 1, bands, 9, 9, generator=g).to(DEV)
    n = (torch.rand(N, bands // 8, generator=g), torch.rand(N, 9, generator=g))
    res = {}
    # "0": layer at a time; "1": fused, forward split into the in-register attention half + row-panel MLP half (default);
    # "1-onekernel": fused, the one-kernel forward of rounds 1-2 (HSIMAE_DEC_SPLIT=0)
    for mode in ("0", "1", "1-onekernel"):
        os.environ["HSIMAE_FUSED_DEC"] = mode[0]
        if mode.endswith("onekernel"):
            os.environ["HSIMAE_DEC_SPLIT"] = "0"
        try:
            m.zero_grad()
            loss, pred, _ = m(x, 0.75, noise=n, grid=grid)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (loss.item(), pred.clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        finally:
            os.environ.pop("HSIMAE_FUSED_DEC", None)
            os.environ.pop("HSIMAE_DEC_SPLIT", None)
    l0, p0, g0 = res["0"]
    for mode in ("1", "1-onekernel"):
        l1, p1, g1 = res[mode]
        print(f"[fused-dec {bands} {mode}] loss layerwise {l0:.7f} fused {l1:.7f}")
        assert abs(l0 - l1) <= 5e-5 * abs(l0)
        assert rms_rel(p1, p0) < 3e-3
        worst = ("", 0.0)
        for k in g0:
            if k.endswith("attn.k.bias"):
                continue
            r = rms_rel(g1[k], g0[k])
            if r > worst[1]:
                worst = (k, r)
            assert r < 3e-2, (k, r)     # two bf16 pipelines, each ~1.5e-2 from the fp32 oracle
        print(f"[fused-dec {bands} {mode}] worst grad rms-rel vs layerwise {worst}")


@pytest.mark.parametrize("bands,grid,N", [(48, (2, 7), 37), (96, (3, 9), 24), (96, (9, 3), 24)])
def test_fused_encoder_mlp_matches_layerwise(bands, grid, N):
    """fused_enc.hip (LN2 -> W1|W3 -> gate -> W2 (+x1) in one kernel, and its recompute backward) against the
    layer-at-a-time kernels on the same inputs; N = 37 makes the 128-row panels ragged."""
    cfg = O.OracleConfig(bands=bands)
    m = build(cfg, O.init_state(cfg, seed=12, std=0.06))
    g = torch.Generator().manual_seed(22)
    x = torch.rand(N, 1, bands, 9, 9, generator=g).to(DEV)
    n = (torch.rand(N, bands // 8, generator=g), torch.rand(N, 9, generator=g))
    res = {}
    for mode in ("0", "1"):
        os.environ["HSIMAE_FUSED_MLP"] = mode
        try:
            m.zero_grad()
            loss, pred, _ = m(x, 0.75, noise=n, grid=grid)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (loss.item(), pred.clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        finally:
            os.environ.pop("HSIMAE_FUSED_MLP", None)
    l0, p0, g0 = res["0"]
    l1, p1, g1 = res["1"]
    print(f"[fused-mlp {bands} {grid}] loss layerwise {l0:.7f} fused {l1:.7f}")
    assert abs(l0 - l1) <= 5e-5 * abs(l0)
    assert rms_rel(p1, p0) < 3e-3
    worst = ("", 0.0)
    for k in g0:
        if k.endswith("attn.k.bias"):
            continue
        r = rms_rel(g1[k], g0[k])
        if r > worst[1]:
            worst = (k, r)
        assert r < 3e-2, (k, r)
    print(f"[fused-mlp {bands}] worst grad rms-rel vs layerwise {worst}")


@pytest.mark.parametrize("N", [1, 3, 5])
def test_tiny_and_odd_batches_against_oracle(N):
    """Batches smaller than any kernel's panel / sample group (one workgroup mostly padding, a last group with one sample):
    loss, mask and a deep weight gradient against the oracle at Base width."""
    cfg = O.OracleConfig(bands=96)
    state = O.init_state(cfg, seed=2, std=0.04)
    m = build(cfg, state)
    g = torch.Generator().manual_seed(N)
    x = torch.rand(N, 1, 96, 9, 9, generator=g)
    n1, n2 = torch.rand(N, 12, generator=g), torch.rand(N, 9, generator=g)
    ref_loss, _, ref_mask, ref_g = O.forward_backward(state, cfg, x, n1.numpy(), n2.numpy(), 3, 9)
    loss, _, mask = m(x.to(DEV), 0.75, noise=(n1, n2), grid=(3, 9))
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(mask.cpu(), ref_mask)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * ref_loss.item()
    named = dict(m.named_parameters())
    for k in ("blocks_1.3.mlp.w1.weight", "blocks_2.0.attn.q.weight", "decoder_blocks.2.mlp.w2.weight", "patch_embed.proj.weight"):
        assert grad_err(named, ref_g, k) < 2e-2, k


@pytest.mark.parametrize("bands,grid", [(64, (3, 6)), (80, (3, 7)), (32, (2, 4)), (16, (2, 2))])
def test_sequence_lengths_with_padded_key_tiles_against_oracle(bands, grid):
    """Decoder sequences of 72 / 90 tokens (on the 7-tile fused kernels: key tiles partly or wholly padding) and 36 / 18 tokens
    (on the 4-tile ones) against the oracle: loss, masks and decoder / encoder gradients.  Rounds 1-2 masked only the last key
    tile in the fused attention backward, which is right for the named configurations (54 and 108 tokens) only."""
    cfg = O.OracleConfig(bands=bands)
    state = O.init_state(cfg, seed=4, std=0.04)
    m = build(cfg, state)
    N = 12
    g = torch.Generator().manual_seed(bands)
    x = torch.rand(N, 1, bands, 9, 9, generator=g)
    n1, n2 = torch.rand(N, bands // 8, generator=g), torch.rand(N, 9, generator=g)
    ref_loss, _, ref_mask, ref_g = O.forward_backward(state, cfg, x, n1.numpy(), n2.numpy(), *grid)
    loss, _, mask = m(x.to(DEV), 0.75, noise=(n1, n2), grid=grid)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(mask.cpu(), ref_mask)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * ref_loss.item()
    named = dict(m.named_parameters())
    for k in ("decoder_blocks.0.norm1.bias", "decoder_blocks.0.attn.q.weight", "decoder_blocks.3.attn.v.weight",
              "decoder_blocks.5.attn.proj.weight", "decoder_blocks.7.mlp.w1.weight", "decoder_embed.weight", "blocks.1.mlp.w2.weight"):
        assert grad_err(named, ref_g, k) < 3e-2, (k, grad_err(named, ref_g, k))      # (18-token sequences: few rows per gradient)


@pytest.mark.parametrize("bands,grid", [(48, (2, 7)), (96, (9, 3))])
def test_public_sub_entry_points_against_oracle(bands, grid):
    """The module's public sub-entry points: forward_encoder -> forward_decoder -> forward_loss -> recons chained
    by hand reproduce the oracle's latent / pred / loss / images, and `mean` / `var` are the target statistics the
    reference stashes (Models.py:537-625)."""
    cfg = O.OracleConfig(bands=bands)
    state = O.init_state(cfg, seed=5, std=0.05)
    m = build(cfg, state).eval()
    N = 12
    g = torch.Generator().manual_seed(31)
    x = torch.rand(N, 1, bands, 9, 9, generator=g)
    n1, n2 = torch.rand(N, bands // 8, generator=g), torch.rand(N, 9, generator=g)
    taps = {}
    ref_loss, ref_pred_img, ref_mask_img = O.forward(state, cfg, x, n1.numpy(), n2.numpy(), *grid, taps)
    xd = x.to(DEV)
    latent, mask, ids_restore, ids_keep = m.forward_encoder(xd, 0.75, noise=(n1, n2), grid=grid)
    assert torch.equal(ids_keep.cpu(), taps["ids_keep"]) and torch.equal(ids_restore.cpu(), taps["ids_restore"])
    assert torch.equal(mask.cpu(), taps["mask"]) and ids_restore.dtype == torch.int64
    assert rms_rel(latent, taps["latent"]) < 5e-3
    pred = m.forward_decoder(latent, ids_restore)
    assert pred.shape == taps["pred"].shape and rms_rel(pred, taps["pred"]) < 1e-2
    loss = m.forward_loss(xd, pred, mask)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * ref_loss.item()
    # the loss kernel alone on the oracle's own prediction: fp32 arithmetic only
    loss2 = m.forward_loss(xd, taps["pred"].to(DEV), mask)
    assert abs(loss2.item() - ref_loss.item()) <= 2e-6 * ref_loss.item()
    tgt = O.patchify(x, cfg)
    assert float((m.mean.cpu() - tgt.mean(-1, keepdim=True)).abs().max()) < 1e-6
    assert float((m.var.cpu() - (tgt.var(-1, keepdim=True) + 1e-6) ** 0.5).abs().max()) < 1e-6
    mask_img, pred_img = m.recons(mask, pred)
    assert torch.equal(mask_img.cpu(), ref_mask_img)
    assert rms_rel(pred_img, ref_pred_img) < 1e-2
    with pytest.raises(ValueError):
        m.forward_decoder(latent[:, :-1], ids_restore)


@pytest.mark.parametrize("bands,grid,N", [(48, (2, 7), 37), (96, (3, 9), 24), (96, (9, 3), 24)])
def test_fused_attention_half_matches_separate_kernels(bands, grid, N):
    """blk128_fwd_kernel (LN1 -> q|k|v -> attention -> projection + residual in one persistent launch) against
    the layer-at-a-time kernels (LN + q|k|v GEMM, attn16_fwd, projection GEMM) on the same inputs: loss, predictions and every gradient (the backward consumes
    the u / qkv / o / lse / x1 the forward saved, so it checks those too)."""
    cfg = O.OracleConfig(bands=bands)
    m = build(cfg, O.init_state(cfg, seed=13, std=0.06))
    g = torch.Generator().manual_seed(23)
    x = torch.rand(N, 1, bands, 9, 9, generator=g).to(DEV)
    n = (torch.rand(N, bands // 8, generator=g), torch.rand(N, 9, generator=g))
    res = {}
    for mode in ("0", "1"):
        os.environ["HSIMAE_FUSED_ATTN_BLOCK"] = mode
        try:
            m.zero_grad()
            loss, pred, _ = m(x, 0.75, noise=n, grid=grid)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (loss.item(), pred.clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        finally:
            os.environ.pop("HSIMAE_FUSED_ATTN_BLOCK", None)
    l0, p0, g0 = res["0"]
    l1, p1, g1 = res["1"]
    print(f"[fused-attn-half {bands} {grid}] loss separate {l0:.7f} fused {l1:.7f}")
    assert abs(l0 - l1) <= 5e-5 * abs(l0)
    assert rms_rel(p1, p0) < 3e-3
    worst = ("", 0.0)
    for k in g0:
        if k.endswith("attn.k.bias"):
            continue
        r = rms_rel(g1[k], g0[k])
        if r > worst[1]:
            worst = (k, r)
        assert r < 3e-2, (k, r)
    print(f"[fused-attn-half {bands}] worst grad rms-rel vs separate {worst}")


@pytest.mark.parametrize("bands,grid,N,det", [(48, (2, 7), 37, False), (96, (3, 9), 24, False), (96, (9, 3), 25, False), (96, (9, 3), 1, False),
                                              (96, (3, 9), 24, True)])
def test_fused_attention_half_backward_matches_separate_kernels(bands, grid, N, det):
    """blk128_bwd_kernel (dO = dx1 Wp -> attention backward -> du = dq|dk|dv Wqkv -> LayerNorm-1 backward + residual, dgamma /
    dbeta, in one persistent launch) against the layer-at-a-time kernels (dO GEMM, attn16_bwd, du GEMM with the LayerNorm backward as its epilogue): every gradient (the q / k / v weight
    gradients read the dq|dk|dv rows the kernel writes, everything upstream reads its dx), both axis-class modes and the
    whole-sample fusion blocks, odd sample counts (pairs of samples per iteration), 14-token sequences (one key tile), the block
    whose dx accumulates into the other stack's, and the deterministic commit path of dgamma / dbeta.  Three schedules:
    separate kernels on the q|k|v the forward saved; the fused kernel on the saved q|k|v; the default — the forward saves no
    q|k|v and the fused kernel recomputes them from u (the same MFMAs on the same operands: the two fused runs must agree to
    the summation order of the gradient commits)."""
    cfg = O.OracleConfig(bands=bands)
    m = build(cfg, O.init_state(cfg, seed=13, std=0.06))
    m.deterministic = det
    g = torch.Generator().manual_seed(23)
    x = torch.rand(N, 1, bands, 9, 9, generator=g).to(DEV)
    n = (torch.rand(N, bands // 8, generator=g), torch.rand(N, 9, generator=g))
    res = {}
    for mode, env in (("separate", {"HSIMAE_FUSED_ATTN_BLOCK_BWD": "0"}), ("saved", {"HSIMAE_ATTN_BWD_RECOMPUTE": "0"}), ("recompute", {})):
        os.environ.update(env)
        try:
            m.zero_grad()
            loss, pred, _ = m(x, 0.75, noise=n, grid=grid)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (loss.item(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        finally:
            for k in env:
                os.environ.pop(k, None)
    (l0, g0), (l1, g1), (l2, g2) = res["separate"], res["saved"], res["recompute"]
    assert l0 == l1 == l2                               # same forward arithmetic
    worst, worst2 = ("", 0.0), ("", 0.0)
    for k in g0:
        assert torch.isfinite(g1[k]).all() and torch.isfinite(g2[k]).all(), k
        if k.endswith("attn.k.bias"):
            continue
        r, r2 = rms_rel(g1[k], g0[k]), rms_rel(g2[k], g1[k])
        worst = max(worst, (k, r), key=lambda t: t[1])
        worst2 = max(worst2, (k, r2), key=lambda t: t[1])
        # (the fused kernel forms delta_i = sum_j P_ij dP_ij in fp32 inside its core, the separate one rowsum(dO * O) from the bf16 O the
        #  forward saved: two roundings of the same quantity; q / k gradients, the most sensitive family, differ by up to 6e-3)
        assert r < 2e-2, (k, r)
        assert r2 < (1e-6 if det else 2e-4), (k, r2)
    print(f"[fused-attn-half-bwd {bands} {grid} N={N} det={det}] worst grad rms-rel fused vs separate {worst}, recompute vs saved {worst2}")


@pytest.mark.parametrize("bands,grid,N", [(96, (3, 9), 21), (96, (9, 3), 21), (48, (2, 7), 9), (96, (3, 9), 1)])
def test_fused_attention_half_d256_matches_separate_kernels(bands, grid, N):
    """blk256_fwd_kernel (attn_wide.hip: LN1 -> q|k|v -> attention -> projection + residual of a D = 256 block in one persistent
    launch, 16 waves = 16 heads, weights streamed) against the layer-at-a-time kernels it replaces (LN1 + q|k|v GEMM, attn16_fwd,
    projection + residual GEMM) on the same inputs, both axis-class modes and the whole-sample fusion blocks, odd sample counts
    (the kernel walks pairs of samples) and 14-token sequences (one key tile): loss, predictions and every gradient (the
    backward consumes the u / qkv / o / lse / x1 the forward saved, so it checks those too)."""
    cfg = O.OracleConfig(bands=bands, embed_dim=256, num_heads=16)
    m = build(cfg, O.init_state(cfg, seed=13, std=0.05))
    g = torch.Generator().manual_seed(29)
    x = torch.rand(N, 1, bands, 9, 9, generator=g).to(DEV)
    n = (torch.rand(N, bands // 8, generator=g), torch.rand(N, 9, generator=g))
    res = {}
    for mode in ("0", "1"):
        os.environ["HSIMAE_FUSED_ATTN_BLOCK256"] = mode
        try:
            m.zero_grad()
            loss, pred, _ = m(x, 0.75, noise=n, grid=grid)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (loss.item(), pred.clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        finally:
            os.environ.pop("HSIMAE_FUSED_ATTN_BLOCK256", None)
    l0, p0, g0 = res["0"]
    l1, p1, g1 = res["1"]
    print(f"[fused-attn-half-256 {bands} {grid} N={N}] loss separate {l0:.7f} fused {l1:.7f}")
    assert abs(l0 - l1) <= 5e-5 * abs(l0)
    assert rms_rel(p1, p0) < 3e-3
    worst = ("", 0.0)
    for k in g0:
        if k.endswith("attn.k.bias"):
            continue
        r = rms_rel(g1[k], g0[k])
        if r > worst[1]:
            worst = (k, r)
        assert r < 3e-2, (k, r)
    print(f"[fused-attn-half-256 {bands}] worst grad rms-rel vs separate {worst}")


@pytest.mark.parametrize("bands,grid,N,det", [(96, (3, 9), 21, False), (96, (9, 3), 22, False), (48, (2, 7), 9, False), (96, (3, 9), 1, False),
                                              (96, (9, 3), 7, True)])
def test_fused_attention_half_backward_d256_matches_separate_kernels(bands, grid, N, det):
    """blk256_bwd_kernel (attn_wide.hip, round 5: dO = dx1 Wp -> attention backward with delta = sum P dP -> du = dq|dk|dv Wqkv ->
    LayerNorm-1 backward + residual gradient, 16 waves = 16 heads, weights streamed) against the three launches it replaces
    (projection data-gradient GEMM, attn16_bwd, q|k|v data-gradient GEMM with the LayerNorm-backward epilogue) on the same
    forward: both axis-class modes and the fusion blocks, odd sample counts (the kernel walks pairs), 14-token sequences (one
    key tile), a single sample, and deterministic mode (where the dgamma / dbeta commits go through the fixed-point path)."""
    cfg = O.OracleConfig(bands=bands, embed_dim=256, num_heads=16)
    m = build(cfg, O.init_state(cfg, seed=13, std=0.05))
    m.deterministic = det
    g = torch.Generator().manual_seed(31)
    x = torch.rand(N, 1, bands, 9, 9, generator=g).to(DEV)
    n = (torch.rand(N, bands // 8, generator=g), torch.rand(N, 9, generator=g))
    res = {}
    for mode in ("0", "1"):
        os.environ["HSIMAE_FUSED_ATTN_BLOCK256_BWD"] = mode
        try:
            m.zero_grad()
            loss, pred, _ = m(x, 0.75, noise=n, grid=grid)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (loss.item(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        finally:
            os.environ.pop("HSIMAE_FUSED_ATTN_BLOCK256_BWD", None)
    (l0, g0), (l1, g1) = res["0"], res["1"]
    assert l0 == l1 or abs(l0 - l1) <= 1e-6 * abs(l0)            # the same forward
    worst = ("", 0.0)
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        if k.endswith("attn.k.bias"):
            continue
        r = rms_rel(g1[k], g0[k])
        if r > worst[1]:
            worst = (k, r)
        assert r < 2e-2, (k, r)
    print(f"[fused-attn-half-256 bwd {bands} {grid} N={N}] worst grad rms-rel vs separate {worst}")


@pytest.mark.parametrize("name,bands,dim,grid,N", [("C3-Large", 96, 256, (3, 9), 12), ("C3-Large", 96, 256, (9, 3), 12),
                                                   ("C5-Huge@512", 192, 512, (6, 9), 6), ("C5-Huge@512", 192, 512, (9, 6), 6),
                                                   ("C5-Huge@512", 192, 512, (18, 3), 6)])
def test_large_and_huge_widths_against_oracle(name, bands, dim, grid, N):
    """Configs C3 (Large, D = 256, 16 heads) and C5 ("Huge" is not defined by the reference: D = 512, 32 heads,
    192 bands) at a small batch against the fp32 oracle: layer-at-a-time encoder kernels at
    K = 256 / 512, 216-token decoder attention (14 key tiles), reference weight scale."""
    cfg = O.OracleConfig(bands=bands, embed_dim=dim, num_heads=dim // 16)
    state = O.init_state(cfg, seed=3, std=0.02)
    g = torch.Generator().manual_seed(77)
    x = torch.rand(N, 1, bands, 9, 9, generator=g)
    n1, n2 = torch.rand(N, bands // 8, generator=g), torch.rand(N, 9, generator=g)
    ref_loss, ref_pred, ref_mask, ref_grads = O.forward_backward(state, cfg, x, n1.numpy(), n2.numpy(), *grid)
    m = build(cfg, state)
    loss, pred, mask = m(x.to(DEV), 0.75, noise=(n1, n2), grid=grid)
    loss.backward()
    torch.cuda.synchronize()
    rel = abs(loss.item() - ref_loss.item()) / ref_loss.item()
    print(f"[{name} {grid}] loss {loss.item():.7f} oracle {ref_loss.item():.7f} rel {rel:.2e}")
    assert torch.equal(mask.cpu(), ref_mask)
    assert rel <= 1e-4
    assert rms_rel(pred, ref_pred) < 1e-2
    named = dict(m.named_parameters())
    worst = ("", 0.0)
    for k in ref_grads:
        r = grad_err(named, ref_grads, k)
        if r > worst[1]:
            worst = (k, r)
        assert r < 3e-2, (k, r)
    print(f"[{name}] worst grad rms-rel {worst}")


def test_large_n64_both_grids_against_oracle():
    """C3's model (HSIMAE-Large, the reference's default `enc_paras = [12, 256, 9]`, Model_Pretraining.py:130) at batch 64
    on both candidate grids: fused MLP-half kernels at D = 256, 64-row GEMM panels, k-outer data-gradient GEMMs."""
    cfg = O.OracleConfig(bands=96, embed_dim=256, num_heads=16)
    state = O.init_state(cfg, seed=11, std=0.02)
    N = 64
    g = torch.Generator().manual_seed(5)
    x = torch.rand(N, 1, 96, 9, 9, generator=g)
    n1, n2 = torch.rand(N, 12, generator=g), torch.rand(N, 9, generator=g)
    for grid in ((3, 9), (9, 3)):
        ref_loss, ref_pred, ref_mask, ref_grads = O.forward_backward(state, cfg, x, n1.numpy(), n2.numpy(), *grid)
        m = build(cfg, state)
        loss, pred, mask = m(x.to(DEV), 0.75, noise=(n1, n2), grid=grid)
        loss.backward()
        assert torch.equal(mask.cpu(), ref_mask)
        rel = abs(loss.item() - ref_loss.item()) / ref_loss.item()
        assert rel <= 1e-4, rel
        named = dict(m.named_parameters())
        worst = max((grad_err(named, ref_grads, k), k) for k in ref_grads)
        assert worst[0] < 3e-2, worst
        print(f"[Large N=64 {grid}] loss rel {rel:.2e}, worst grad rms-rel {worst}")


@pytest.mark.parametrize("name,bands,dim,N,prec", [("Large", 96, 256, 4096, "bf16"), ("Huge fp8", 192, 512, 1024, "fp8")])
def test_large_and_huge_full_size_properties(name, bands, dim, N, prec):
    """BASELINE.json configs[2] / configs[4] at their per-GPU batch (4096 / 1024): size-independent properties — finite loss
    and gradients, masks bit-exact against the oracle's closed form, the kept-token count, and batch-size independence of
    the per-sample predictions (the first 32 cubes alone give the same prediction images)."""
    cfg = O.OracleConfig(bands=bands, embed_dim=dim, num_heads=dim // 16)
    m = build(cfg, O.init_state(cfg, seed=5, std=0.02)).set_precision(prec)
    T = bands // 8
    torch.manual_seed(7)
    x = torch.rand(N, 1, bands, 9, 9, device=DEV)
    n1, n2 = torch.rand(N, T), torch.rand(N, 9)
    cands = HSIMAE.grid_candidates(T, 9, 0.75)            # every grid bench.py draws (round 4 ran the first one only)
    assert len(cands) >= 2
    for grid in cands:
        K = grid[0] * grid[1]
        m.zero_grad(set_to_none=True)
        loss, pred, mask = m(x, 0.75, noise=(n1, n2), grid=grid)
        loss.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(loss) and torch.isfinite(pred).all() and 0.5 < loss.item() < 2.0, grid
        assert float(mask.sum()) == N * (T * 9 - K) * 72
        k2, r2, m2 = O.mask_from_noise(n1.numpy(), n2.numpy(), *grid)
        assert torch.equal(mask.cpu(), O.unpatchify(torch.from_numpy(m2).unsqueeze(2).repeat(1, 1, 72), cfg))
        for pname, p in m.named_parameters():
            if p.requires_grad and pname != "mask_token":
                assert p.grad is not None and torch.isfinite(p.grad).all(), (grid, pname)
        with torch.no_grad():
            small = m(x[:32], 0.75, noise=(n1[:32], n2[:32]), grid=grid)[1]
        assert rms_rel(small, pred[:32]) < 1e-6, grid


def test_fused_adamw_matches_torch_adamw_and_training_step():
    """hsimae_adamw_step (one launch over the flat buffer) against torch.optim.AdamW with the reference's two
    name-filtered groups (Model_Pretraining.py:80-86), 6 steps on the same gradients; then a real training loop."""
    from hsimae_amd import FusedAdamW
    cfg = O.OracleConfig(bands=48)
    state = O.init_state(cfg, seed=9, std=0.02)
    ma, mb = build(cfg, state), build(cfg, state)
    g = torch.Generator().manual_seed(31)
    x = torch.rand(16, 1, 48, 9, 9, generator=g).to(DEV)
    nd = ["bias", "norm"]
    groups = [{"params": [p for n, p in ma.named_parameters() if not any(k in n for k in nd)], "weight_decay": 5e-2},
              {"params": [p for n, p in ma.named_parameters() if any(k in n for k in nd)], "weight_decay": 0.0}]
    ref = torch.optim.AdamW(groups, lr=5e-3, weight_decay=5e-2, betas=(0.9, 0.95))
    fused = FusedAdamW(mb, lr=5e-3, weight_decay=5e-2, betas=(0.9, 0.95))
    la, lb = [], []
    for step in range(6):
        n = (torch.rand(16, 6, generator=g), torch.rand(16, 9, generator=g))
        for m_, opt, acc in ((ma, ref, la), (mb, fused, lb)):
            loss, _, _ = m_(x, 0.75, noise=n, grid=(2, 7))
            opt.zero_grad()
            loss.backward()
            opt.step()
            acc.append(loss.item())
    print("[adamw] torch", [f"{v:.5f}" for v in la])
    print("[adamw] fused", [f"{v:.5f}" for v in lb])
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-3 * abs(a)          # two independent bf16 runs drift apart through the atomics' order
    worst = 0.0
    for (pname, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        if pname in ("pos_embed", "decoder_pos_embed", "mask_token"):
            assert torch.equal(pa, pb)                # frozen / unused: untouched
            continue
        if pname.endswith("attn.k.bias"):
            continue      # true gradient is exactly zero (softmax shift invariance): Adam turns rounding noise into +-lr
        worst = max(worst, rms_rel(pb, pa))
    print(f"[adamw] worst parameter rms-rel after 6 steps {worst:.2e}")
    assert worst < 0.15       # chaotic bound only (two independent runs, Adam amplifies last-bit gradient noise of
    #                           near-zero gradients to +-lr): measured 2e-2..5e-2; the arithmetic itself is pinned to 1e-6 below
    # exactness of the arithmetic itself: identical gradients in, one step
    mc, md = build(cfg, state), build(cfg, state)
    for m_ in (mc, md):
        m_(x, 0.75, noise=n, grid=(2, 7))[0].backward()
    md._flat_grad.copy_(mc._flat_grad)
    groups = [{"params": [p for n_, p in mc.named_parameters() if not any(k in n_ for k in nd)], "weight_decay": 5e-2},
              {"params": [p for n_, p in mc.named_parameters() if any(k in n_ for k in nd)], "weight_decay": 0.0}]
    torch.optim.AdamW(groups, lr=5e-3, weight_decay=5e-2, betas=(0.9, 0.95)).step()
    FusedAdamW(md, lr=5e-3, weight_decay=5e-2, betas=(0.9, 0.95)).step()
    torch.cuda.synchronize()
    assert float((mc._flat - md._flat).abs().max()) <= 1e-6
