"""Full-size parity against the oracle (VERDICT r04 "Next round" 2): loss AND gradients at the benchmark batch.

The weight-gradient reductions that only exist at full size — 110,592- and 442,368-row contractions, `msplit` row slices meeting in
fp32 atomics, the decoder's 256-workgroup slabs + fixed-order reduce, the two-stream interleaving — were covered by "finite" only
(`test_c2_full_size_properties_n4096`); the largest weight-gradient kernel test stops at M = 13,824.  Here the whole C2 step
(HSIMAE-Base, 9x9x96, N = 4096, both grid candidates) and the Large step at N = 4096 are compared with the CPU oracle.

The oracle runs the batch in chunks of 256 cubes: cubes are independent through the whole forward (Models.py:627-634) and every
cube masks the same number of patches (Models.py:495-535), so the batch loss is the mean of the chunk losses and every gradient
the mean of the chunk gradients — exactly, in fp32 up to summation order.  ~20 s (Base) / ~60 s (Large) of host time per grid.

Gates (stated by the review): loss <= 1e-4 relative; named tensors of every family RMS-relative <= 1e-2 and L2 norm <= 5e-3;
every other trainable tensor RMS-relative <= 2e-2 (the reference-scale figures of profiles/r04_grad_error_c1.txt: median
3.7e-3, worst 7e-3 at N = 64)."""
import os

import pytest
import torch

from hsimae_amd import HSIMAE
from oracle import hsimae_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
DEV = "cuda:0"

NAMED_BASE = ["blocks_1.0.attn.q.weight", "blocks_1.4.attn.proj.weight", "blocks_1.0.norm1.weight", "blocks_1.0.norm1.bias",
              "blocks_2.8.mlp.w2.weight", "blocks_2.3.mlp.w1.weight", "blocks_2.8.mlp.w1.bias", "blocks_2.5.attn.k.weight",
              "blocks.2.mlp.w1.weight", "blocks.0.attn.v.weight", "blocks.1.mlp.w3.weight", "blocks.2.norm2.weight",
              "decoder_blocks.0.attn.q.weight", "decoder_blocks.0.mlp.w1.weight", "decoder_blocks.0.attn.v.bias",
              "decoder_blocks.7.attn.proj.weight", "decoder_blocks.7.mlp.w2.weight", "decoder_blocks.3.norm2.weight",
              "decoder_blocks.3.norm2.bias", "patch_embed.proj.weight", "decoder_pred.weight", "decoder_embed.weight",
              "norm.weight", "decoder_norm.weight"]
NAMED_LARGE = ["blocks_1.0.attn.q.weight", "blocks_2.8.mlp.w2.weight", "blocks.2.mlp.w1.weight", "blocks_2.4.attn.proj.weight",
               "decoder_blocks.0.attn.k.weight", "decoder_blocks.7.mlp.w3.weight", "patch_embed.proj.weight", "norm.weight"]


def oracle_in_chunks(state, cfg, x, n1, n2, grid, chunk=256):
    N = x.shape[0]
    assert N % chunk == 0
    nchunk = N // chunk
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    loss, grads = 0.0, None
    for c in range(nchunk):
        sl = slice(c * chunk, (c + 1) * chunk)
        l, _, _, g = O.forward_backward(state, cfg, x[sl], n1[sl].numpy(), n2[sl].numpy(), *grid)
        loss += float(l) / nchunk
        if grads is None:
            grads = {k: v.double() / nchunk for k, v in g.items()}
        else:
            for k, v in g.items():
                grads[k] += v.double() / nchunk
    return loss, grads


def compare(m, loss, ref_loss, ref_grads, named_keys, tag):
    rel = abs(loss - ref_loss) / ref_loss
    named = dict(m.named_parameters())
    rows, worst_other = [], ("", 0.0)
    for k, ref in ref_grads.items():
        g = named[k].grad
        assert g is not None and torch.isfinite(g).all(), k
        gd = g.double().cpu()
        if k.endswith("attn.k.bias"):           # true gradient exactly zero: the oracle holds rounding noise only
            scale = ref_grads[k.replace(".k.bias", ".q.bias")].pow(2).mean().sqrt()
            assert float((gd - ref).pow(2).mean().sqrt() / scale) <= 5e-2, k
            continue
        rms = float((gd - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30))
        nrm = abs(float(gd.norm()) - float(ref.norm())) / max(float(ref.norm()), 1e-30)
        if k in named_keys:
            rows.append((k, rms, nrm))
        elif rms > worst_other[1]:
            worst_other = (k, rms)
    rows.sort(key=lambda r: -r[1])
    print(f"[{tag}] loss {loss:.7f} vs oracle {ref_loss:.7f} (rel {rel:.2e}); named tensors worst rms-rel "
          f"{rows[0][0]} {rows[0][1]:.2e}, worst norm err {max(r[2] for r in rows):.2e}; other tensors worst {worst_other[0]} {worst_other[1]:.2e}")
    assert len(rows) == len(named_keys), set(named_keys) - {r[0] for r in rows}
    assert rel <= 1e-4, f"{tag}: loss {loss} vs oracle {ref_loss} (rel {rel:.2e})"
    for k, rms, nrm in rows:
        assert rms <= 1e-2 and nrm <= 5e-3, f"{tag}: {k} rms-rel {rms:.2e}, norm err {nrm:.2e}"
    assert worst_other[1] <= 2e-2, f"{tag}: {worst_other}"


@pytest.mark.parametrize("grid", [(3, 9), (9, 3)])
def test_c2_n4096_loss_and_gradients_against_the_oracle(grid):
    cfg = O.OracleConfig(bands=96)
    state = O.init_state(cfg, seed=5, std=0.02)         # the reference's weight scale (Models.py:452)
    N = 4096
    g = torch.Generator().manual_seed(7)
    x = torch.rand(N, 1, 96, 9, 9, generator=g)
    n1, n2 = torch.rand(N, 12, generator=g), torch.rand(N, 9, generator=g)
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, embed_dim=128, depth=12, num_heads=8, s_depth=9,
               decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True)
    m.load_state_dict(state)
    m = m.to(DEV)
    loss, _, mask = m(x.to(DEV), 0.75, noise=(n1, n2), grid=grid)
    loss.backward()
    torch.cuda.synchronize()
    ref_loss, ref_grads = oracle_in_chunks(state, cfg, x, n1, n2, grid)
    compare(m, loss.item(), ref_loss, ref_grads, NAMED_BASE, f"C2 N=4096 grid {grid}")


def test_large_n4096_loss_and_gradients_against_the_oracle():
    cfg = O.OracleConfig(bands=96, embed_dim=256, num_heads=16)
    state = O.init_state(cfg, seed=5, std=0.02)
    N = 4096
    g = torch.Generator().manual_seed(9)
    x = torch.rand(N, 1, 96, 9, 9, generator=g)
    n1, n2 = torch.rand(N, 12, generator=g), torch.rand(N, 9, generator=g)
    grid = (9, 3)
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, embed_dim=256, depth=12, num_heads=16, s_depth=9,
               decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True)
    m.load_state_dict(state)
    m = m.to(DEV)
    loss, _, _ = m(x.to(DEV), 0.75, noise=(n1, n2), grid=grid)
    loss.backward()
    torch.cuda.synchronize()
    ref_loss, ref_grads = oracle_in_chunks(state, cfg, x, n1, n2, grid)
    compare(m, loss.item(), ref_loss, ref_grads, NAMED_LARGE, f"Large N=4096 grid {grid}")
