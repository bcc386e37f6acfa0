#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE implementation.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  It *imports* the reference's `Models.py`, replays its RNG streams and records the
reference's own inputs/outputs as data.  No reference source text is copied.

    python tests/golden/make_golden.py

Fixtures:
  manifest.json          F1  state_dict (key, shape, dtype) manifests: HSIMAE C1/C2/C3, DualViT, HSIViT
  masking.npz/.json      F2  noise -> ids_keep / ids_restore / mask, candidate grids, python-random draws
  tiny_model.npz         F3  tiny model: state_dict, inputs, noise, every stage output, all parameter grads
  pos_embed.npz          F4  sin-cos tables
  c1_summary.json/.npz   F5  Base/48 bands: loss fp32/fp64, stage checksums, grad norms (non-degenerate weights)
  trajectory.json        F6  10 AdamW steps (small config), losses
"""
import contextlib
import io
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
with contextlib.redirect_stdout(io.StringIO()):
    import Models as R  # noqa: E402  (the reference, imported in place)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_mae(bands, dim, dec_dim=64, depth=12, s_depth=9, dec_depth=8, heads=None, dec_heads=None, norm_pix=True):
    return quiet(R.HSIMAE, img_size=9, patch_size=3, in_chans=1, bands=bands, b_patch_size=8,
                 embed_dim=dim, depth=depth, num_heads=heads or dim // 16, s_depth=s_depth,
                 decoder_embed_dim=dec_dim, decoder_depth=dec_depth, decoder_num_heads=dec_heads or dec_dim // 8,
                 norm_pix_loss=norm_pix, trunc_init=True)


def manifest(sd):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]


def replay(model, x, ratio):
    """Run the reference forward once and return what its RNG streams produced."""
    T, L = model.input_size[0], model.input_size[1] ** 2
    st, pst = torch.get_rng_state(), random.getstate()
    out = model(x, ratio)
    after_t, after_p = torch.get_rng_state(), random.getstate()
    torch.set_rng_state(st); random.setstate(pst)
    cands = torch.tensor([(a, b) for a in range(2, T + 1) for b in range(2, L + 1)])
    len_keep = (1 - ratio) * T * L
    diff = abs(len_keep - cands[:, 0] * cands[:, 1])
    ind = torch.where(diff == diff.min())[0]
    j = random.sample(range(len(ind)), 1)[0]
    n1 = torch.rand(x.shape[0], T)
    n2 = torch.rand(x.shape[0], L)
    assert torch.equal(torch.get_rng_state(), after_t) and random.getstate() == after_p, "RNG replay mismatch"
    torch.set_rng_state(after_t); random.setstate(after_p)
    assert (int(model.len_t), int(model.len_l)) == tuple(cands[ind[j]].tolist())
    return out, n1, n2, int(model.len_t), int(model.len_l), [tuple(c) for c in cands[ind].tolist()]


def perturb(model, seed, std=0.2):
    """Non-degenerate weights: at trunc_init the loss barely depends on the network."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
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


def f1_manifest():
    out = {}
    for name, (bands, dim) in {"C1_base48": (48, 128), "C2_base96": (96, 128), "C3_large96": (96, 256)}.items():
        out[name] = manifest(make_mae(bands, dim).state_dict())
    dv = quiet(R.DualViT, img_size=9, patch_size=3, in_chans=1, bands=32, b_patch_size=8, num_class=16,
               embed_dim=128, depth=12, num_heads=8, s_depth=9, decoder_embed_dim=64, decoder_depth=8,
               decoder_num_heads=8, norm_pix_loss=True, trunc_init=True, drop_path=0.2)
    out["DualViT_base32"] = manifest(dv.state_dict())
    out["HSIMAE_base32"] = manifest(make_mae(32, 128).state_dict())
    vit = quiet(R.HSIViT, img_size=9, patch_size=3, in_chans=1, bands=32, b_patch_size=8, num_class=16,
                embed_dim=128, depth=12, num_heads=8, s_depth=9)
    out["HSIViT_base32"] = manifest(vit.state_dict())
    # param-group split used by Model_Pretraining.py:80-84
    m = make_mae(96, 128)
    nd = ["bias", "norm"]
    names = [n for n, _ in m.named_parameters()]
    out["named_parameters_C2"] = names
    out["no_decay_count_C2"] = sum(any(k in n for k in nd) for n in names)
    out["trainable_numel"] = {k: int(sum(p.numel() for p in make_mae(b, d).parameters() if p.requires_grad))
                              for k, (b, d) in {"C1_base48": (48, 128), "C3_large96": (96, 256)}.items()}
    out["versions"] = {"torch": torch.__version__, "numpy": np.__version__}
    json.dump(out, open(os.path.join(HERE, "manifest.json"), "w"))


def f2_masking():
    arrs, meta = {}, {"cases": [], "draws": {}}
    model = make_mae(48, 32, dec_dim=16, depth=1, s_depth=0, dec_depth=1, heads=2, dec_heads=2)
    for ci, (T, L, r) in enumerate([(6, 9, .75), (12, 9, .75), (24, 9, .75), (4, 9, .5), (4, 9, .8), (12, 9, .5)]):
        for seed in (0, 1, 2):
            torch.manual_seed(100 + seed); random.seed(seed)
            N = 16
            st, pst = torch.get_rng_state(), random.getstate()
            x = torch.zeros(N, T * L, 4)
            _, mask, ids_restore, ids_keep = model.spatial_spectral_masking(x, T, L, r)
            torch.set_rng_state(st); random.setstate(pst)
            cands = R.HSIMAE.get_dim_patches  # noqa (only to document the source)
            pairs = [(a, b) for a in range(2, T + 1) for b in range(2, L + 1)]
            lens = torch.tensor([a * b for a, b in pairs])
            diff = abs((1 - r) * T * L - lens)
            ind = torch.where(diff == diff.min())[0].tolist()
            j = random.sample(range(len(ind)), 1)[0]
            n1 = torch.rand(N, T); n2 = torch.rand(N, L)
            key = f"c{ci}_s{seed}"
            arrs[key + "_n1"] = n1.numpy(); arrs[key + "_n2"] = n2.numpy()
            arrs[key + "_keep"] = ids_keep.numpy().astype(np.int16)
            arrs[key + "_restore"] = ids_restore.numpy().astype(np.int16)
            arrs[key + "_mask"] = mask.numpy().astype(np.uint8)
            meta["cases"].append({"key": key, "T": T, "L": L, "ratio": r, "seed": seed,
                                  "len_t": int(model.len_t), "len_l": int(model.len_l),
                                  "candidates": [list(pairs[i]) for i in ind], "draw": j})
    # python-random draw sequence for seeds 0..9 (5 successive forwards each), T=12 L=9 r=.75
    for seed in range(10):
        random.seed(seed)
        seq = []
        for _ in range(5):
            t, l = model.get_dim_patches(12, 9, 0.75)
            seq.append([int(t), int(l)])
        meta["draws"][str(seed)] = seq
    np.savez_compressed(os.path.join(HERE, "masking.npz"), **arrs)
    json.dump(meta, open(os.path.join(HERE, "masking.json"), "w"))


def taps_forward(model, x, ratio):
    taps = {}
    hooks = []

    def hook(name):
        return lambda mod, inp, out: taps.__setitem__(name, out.detach().clone())

    hooks.append(model.patch_embed.register_forward_hook(hook("patch_embed")))
    if hasattr(model, "blocks_1"):
        hooks.append(model.blocks_1[-1].register_forward_hook(hook("x1_seq")))
        hooks.append(model.blocks_2[-1].register_forward_hook(hook("x2_seq")))
        hooks.append(model.blocks_1[0].register_forward_pre_hook(
            lambda mod, inp: taps.__setitem__("enc_in_seq", inp[0].detach().clone())))
    if hasattr(model, "blocks"):
        hooks.append(model.blocks[-1].register_forward_hook(hook("fused")))
    hooks.append(model.norm.register_forward_hook(hook("latent")))
    hooks.append(model.decoder_blocks[0].register_forward_pre_hook(
        lambda mod, inp: taps.__setitem__("dec_in", inp[0].detach().clone())))
    hooks.append(model.decoder_blocks[-1].register_forward_hook(hook("dec_out")))
    hooks.append(model.decoder_pred.register_forward_hook(hook("pred")))
    orig = model.spatial_spectral_masking

    def wrapped(xx, T, L, r):
        o = orig(xx, T, L, r)
        taps["mask"], taps["ids_restore"], taps["ids_keep"] = o[1].clone(), o[2].clone(), o[3].clone()
        return o

    model.spatial_spectral_masking = wrapped
    out, n1, n2, lt, ll, cands = replay(model, x, ratio)
    model.spatial_spectral_masking = orig
    for h in hooks:
        h.remove()
    return out, taps, n1, n2, lt, ll, cands


def f3_tiny():
    torch.manual_seed(7); random.seed(7)
    model = make_mae(32, 32, dec_dim=32, depth=3, s_depth=2, dec_depth=2, heads=2, dec_heads=4)
    perturb(model, 11)
    N = 6
    x = torch.rand(N, 1, 32, 9, 9)
    for ratio, tag in ((0.5, "r50"), (0.75, "r75")):
        model.zero_grad()
        (loss, pred, mask), taps, n1, n2, lt, ll, cands = taps_forward(model, x, ratio)
        loss.backward()
        arrs = {"x": x.numpy(), "noise_1": n1.numpy(), "noise_2": n2.numpy(),
                "len_tl": np.array([lt, ll]), "loss": np.array(loss.item(), dtype=np.float64),
                "pred_img": pred.detach().numpy(), "mask_img": mask.numpy().astype(np.uint8),
                "target_mean": model.mean.numpy(), "target_std": model.var.numpy()}
        for k, v in taps.items():
            arrs["tap_" + k] = v.numpy() if v.dtype != torch.int64 else v.numpy().astype(np.int16)
        for k, v in model.state_dict().items():
            arrs["sd_" + k] = v.numpy()
        for k, p in model.named_parameters():
            if p.grad is not None:
                arrs["grad_" + k] = p.grad.numpy()
        arrs["no_grad_names"] = np.array([k for k, p in model.named_parameters() if p.grad is None])
        # strided (band-fastest) input layout as produced by HSIdataset4PT (Model_Pretraining.py:49-50)
        xs = x[:, 0].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).unsqueeze(1)
        assert not xs.is_contiguous() and torch.equal(xs, x)
        l2 = model(xs, ratio)[0] if False else None  # (layout equivalence is value-level; checked in tests)
        np.savez_compressed(os.path.join(HERE, f"tiny_model_{tag}.npz"), **arrs)


def f4_pos():
    arrs = {}
    for D, T in [(128, 6), (128, 12), (256, 12), (64, 12), (64, 24), (32, 4), (16, 4)]:
        arrs[f"D{D}_T{T}"] = R.get_3d_sincos_pos_embed(D, T, 3)[0].numpy()
    np.savez_compressed(os.path.join(HERE, "pos_embed.npz"), **arrs)


def stats(t):
    t = t.double()
    return [float(t.sum()), float(t.abs().sum()), float(t.pow(2).sum().sqrt())]


def f5_c1():
    torch.manual_seed(0); random.seed(0)
    model = make_mae(48, 128)
    perturb(model, 5)
    N = 16
    torch.manual_seed(1234)
    x = torch.rand(N, 1, 48, 9, 9)
    torch.manual_seed(99); random.seed(0)
    (loss, pred, mask), taps, n1, n2, lt, ll, cands = taps_forward(model, x, 0.75)
    loss.backward()
    out = {"N": N, "bands": 48, "len_t": lt, "len_l": ll, "candidates": [list(c) for c in cands],
           "loss_fp32": float(loss.item()), "perturb_seed": 5, "x_seed": 1234,
           "taps": {k: stats(v) for k, v in taps.items() if v.dtype != torch.int64},
           "pred_img": stats(pred), "mask_img_sum": float(mask.sum()),
           "grad_l2": {k: float(p.grad.double().norm()) for k, p in model.named_parameters() if p.grad is not None}}
    # fp64 reference loss, same noise
    m64 = make_mae(48, 128).double()
    m64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
    torch.manual_seed(99); random.seed(0)
    torch.set_default_dtype(torch.float64)
    try:
        # torch.rand in fp64 consumes the generator differently -> inject the fp32 noise instead
        orig_rand = torch.rand
        seq = [n1.double(), n2.double()]
        torch.rand = lambda *a, **k: seq.pop(0)
        l64 = m64(x.double(), 0.75)[0]
    finally:
        torch.rand = orig_rand
        torch.set_default_dtype(torch.float32)
    out["loss_fp64"] = float(l64.item())
    json.dump(out, open(os.path.join(HERE, "c1_summary.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "c1_summary.npz"), noise_1=n1.numpy(), noise_2=n2.numpy(),
                        ids_keep=taps["ids_keep"].numpy().astype(np.int16),
                        latent=taps["latent"].numpy().astype(np.float32)[:4],
                        pred=taps["pred"].numpy().astype(np.float32)[:2])


def f6_traj():
    torch.manual_seed(3); random.seed(3)
    model = make_mae(32, 32, dec_dim=32, depth=3, s_depth=2, dec_depth=2, heads=2, dec_heads=4)
    perturb(model, 21)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    N = 8
    torch.manual_seed(4321)
    x = torch.rand(N, 1, 32, 9, 9)
    nd = ["bias", "norm"]
    groups = [{"params": [p for n, p in model.named_parameters() if not any(k in n for k in nd)], "weight_decay": 5e-2},
              {"params": [p for n, p in model.named_parameters() if any(k in n for k in nd)], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=1e-3, weight_decay=5e-2, betas=(0.9, 0.95))
    torch.manual_seed(55); random.seed(55)
    losses, noises, grids = [], [], []
    for step in range(10):
        (loss, _, _), n1, n2, lt, ll, _ = replay(model, x, 0.5)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss.item())); noises.append((n1.numpy(), n2.numpy())); grids.append([lt, ll])
    arrs = {"x": x.numpy()}
    for k, v in sd0.items():
        arrs["sd_" + k] = v.numpy()
    for i, (a, b) in enumerate(noises):
        arrs[f"n1_{i}"] = a; arrs[f"n2_{i}"] = b
    np.savez_compressed(os.path.join(HERE, "trajectory.npz"), **arrs)
    json.dump({"losses": losses, "grids": grids, "lr": 1e-3, "wd": 5e-2, "betas": [0.9, 0.95], "ratio": 0.5,
               "cfg": {"bands": 32, "embed_dim": 32, "decoder_embed_dim": 32, "depth": 3, "s_depth": 2,
                       "decoder_depth": 2, "num_heads": 2, "decoder_num_heads": 4}},
              open(os.path.join(HERE, "trajectory.json"), "w"))


def f7_init():
    """Parameter values right after construction under a fixed seed (init RNG order, Models.py:429-459)."""
    out = {}
    for tag, kw in {"trunc": dict(trunc_init=True), "xavier": dict(trunc_init=False)}.items():
        torch.manual_seed(0)
        m = quiet(R.HSIMAE, img_size=9, patch_size=3, in_chans=1, bands=48, b_patch_size=8, embed_dim=128, depth=12,
                  num_heads=8, s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8,
                  norm_pix_loss=True, **kw)
        out[tag] = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in m.state_dict().items()}
    json.dump(out, open(os.path.join(HERE, "init_checksums.json"), "w"))


if __name__ == "__main__":
    torch.set_num_threads(8)
    f7_init(); print("F7 ok")
    f1_manifest(); print("F1 ok")
    f2_masking(); print("F2 ok")
    f3_tiny(); print("F3 ok")
    f4_pos(); print("F4 ok")
    f5_c1(); print("F5 ok")
    f6_traj(); print("F6 ok")
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE))
    print("fixture bytes:", tot)
