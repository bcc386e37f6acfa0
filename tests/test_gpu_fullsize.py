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


def oracle_in_chunks(state, cfg, x, n1, n2, grid, chunk=256, operands=None):
    """`operands`: a context-manager factory of the oracle (O.operands_mx8: the encoder linears' operands quantised as the fp8
    kernels quantise them) entered around every chunk; None = the fp32 reference arithmetic."""
    import contextlib
    N = x.shape[0]
    assert N % chunk == 0
    nchunk = N // chunk
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    loss, grads = 0.0, None
    for c in range(nchunk):
        sl = slice(c * chunk, (c + 1) * chunk)
        with (operands() if operands else contextlib.nullcontext()):
            l, _, _, g = O.forward_backward(state, cfg, x[sl], n1[sl].numpy(), n2[sl].numpy(), *grid)
        loss += float(l) / nchunk
        if grads is None:
            grads = {k: v.double() / nchunk for k, v in g.items()}
        else:
            for k, v in g.items():
                grads[k] += v.double() / nchunk
    return loss, grads


def compare(m, loss, ref_loss, ref_grads, named_keys, tag, loss_gate=1e-4, named_gate=1e-2, norm_gate=5e-3, other_gate=2e-2):
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
    assert rel <= loss_gate, f"{tag}: loss {loss} vs oracle {ref_loss} (rel {rel:.2e})"
    for k, rms, nrm in rows:
        assert rms <= named_gate and nrm <= norm_gate, f"{tag}: {k} rms-rel {rms:.2e}, norm err {nrm:.2e}"
    assert worst_other[1] <= other_gate, f"{tag}: {worst_other}"


# std = 0.02 is the reference's weight scale (Models.py:452), where the loss barely depends on the network (~1.01); std = 0.08 is the
# non-degenerate regime (VERDICT r05 weak spot 1 / "Next round" 4a): activations that are not LayerNorm noise through the
# full-size weight-gradient reductions.  Its loss gate is 1e-3 (rounding the WEIGHTS to bf16 alone moves the loss by 1-4e-4 there:
# test_c1_base48_against_oracle_loss_latent_grads), the gradient gates are the same.  Measured (profiles/r06_d_new_parity_tests.txt):
# loss 1.07e-4, named tensors worst 9.3e-3 RMS-rel (a LayerNorm weight) / 3.4e-3 on the norm, every other tensor <= 1.33e-2.
@pytest.mark.parametrize("grid,std,loss_gate", [((3, 9), 0.02, 1e-4), ((9, 3), 0.02, 1e-4), ((9, 3), 0.08, 1e-3)])
def test_c2_n4096_loss_and_gradients_against_the_oracle(grid, std, loss_gate):
    cfg = O.OracleConfig(bands=96)
    state = O.init_state(cfg, seed=5, std=std)
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
    compare(m, loss.item(), ref_loss, ref_grads, NAMED_BASE, f"C2 N=4096 grid {grid} std {std}", loss_gate=loss_gate)


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


NAMED_HUGE = ["blocks_1.0.attn.q.weight", "blocks_1.8.attn.proj.weight", "blocks_2.4.mlp.w1.weight", "blocks_2.8.mlp.w2.weight",
              "blocks.2.mlp.w3.weight", "blocks.0.attn.v.weight", "decoder_blocks.0.mlp.w1.weight", "decoder_blocks.7.attn.proj.weight"]


def test_huge_fp8_n1024_against_the_mx_operand_oracle():
    """VERDICT r05 weak spot 2 / "Next round" 4b: Huge (D = 512, 192 bands) at its benchmark batch was "finite + properties"; the oracle
    comparisons at D = 512 stopped at N = 6.  Here the fp8 step at N = 1024, grid (6, 9): loss and 8 named tensors against the oracle
    run with the SAME operand format (`oracle.operands_mx8`, walked in chunks of 64 cubes) — against the fp32 oracle the distance
    would be the e4m3 operand rounding (4.6 % median), which `test_huge_fp8_error_is_the_mx_operand_rounding` prices at N = 6."""
    cfg = O.OracleConfig(bands=192, embed_dim=512, num_heads=32)
    state = O.init_state(cfg, seed=0, std=0.02)
    N, grid = 1024, (6, 9)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(N, 1, 192, 9, 9, generator=g)
    n1, n2 = torch.rand(N, 24, generator=g), torch.rand(N, 9, generator=g)
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=192, b_patch_size=8, embed_dim=512, depth=12, num_heads=32, s_depth=9,
               decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True)
    m.load_state_dict(state)
    m = m.to(DEV).set_precision("fp8")
    loss, _, _ = m(x.to(DEV), 0.75, noise=(n1, n2), grid=grid)
    loss.backward()
    torch.cuda.synchronize()
    ref_loss, ref_grads = oracle_in_chunks(state, cfg, x, n1, n2, grid, chunk=64, operands=O.operands_mx8)
    # measured (profiles/r06_d_new_parity_tests.txt): loss 1.4e-7, named tensors worst 4.8e-4 RMS-rel / 1.8e-4 on the norm, every other
    # tensor <= 2.6e-3 (a decoder q weight) — gated at ~5x that
    compare(m, loss.item(), ref_loss, ref_grads, NAMED_HUGE, f"Huge fp8 N=1024 grid {grid} vs MX-operand oracle",
            loss_gate=2e-5, named_gate=3e-3, norm_gate=1e-3, other_gate=1.5e-2)
