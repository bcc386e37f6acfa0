"""The fp8 MFMA path (BASELINE.json configs[4], `hsimae_config.precision = FP8`) on a real MI355X.

Kernel level: the MX block-scaled GEMM (v_mfma_scale_f32_16x16x128_f8f6f4, OCP e4m3 operands, one e8m0 scale per 32
consecutive K elements) against a torch emulation of exactly that quantisation — same blocks, same scale rule, same
round-to-nearest-even e4m3 — evaluated in fp64: only the accumulation differs (tolerance 1e-4 of the largest output).
Model level: HSIMAE at the Huge width (embed_dim 512, this repo's definition of "Huge") with the encoder
linears in fp8 against the fp32 CPU oracle: masks / ids bit-exact, loss within the STATED fp8 tolerance of 2e-3 relative
(measured ~1e-4 at the reference's weight scale), gradients RMS-relative <= 0.12 (e4m3 carries 3 mantissa bits: ~3 % per
operand element; weight gradients themselves stay bf16 x bf16).
"""
import contextlib
import ctypes as C
import io

import pytest
import torch

from hsimae_amd import HSIMAE, _lib
from oracle import hsimae_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def stream():
    return torch.cuda.current_stream().cuda_stream


def rup(x, m):
    return (x + m - 1) // m * m


def mx_e4m3(x):
    """MX quantisation along the last dim (32-element blocks, e8m0 scale 2^(floor(log2 amax) - 8), saturating e4m3, RNE),
    returned de-quantised in fp32.  The last dim must be a multiple of 32."""
    sh = x.shape
    b = x.float().reshape(-1, sh[-1] // 32, 32)
    am = b.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(am.clamp_min(2.0 ** -120))) - 8
    e = e.clamp(-126, 127)
    scale = torch.exp2(e)
    q = (b / scale).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    return (q * scale).reshape(sh)


def pack8(W, N_img, K):
    """fp32 W [N, K] -> (e4m3 image, scale image) through hsimae_pack_matrix with desc.fp8 = 1."""
    lib = _lib.load()
    KS = (K + 127) // 128
    img = torch.zeros((N_img // 16) * KS * 64 * 32, dtype=torch.uint8, device=DEV)
    sc = torch.zeros((N_img // 16) * ((KS + 3) // 4) * 64 * 4, dtype=torch.uint8, device=DEV)
    W = W.contiguous().float()
    d = (_lib.PackDesc * 1)()
    d[0] = _lib.PackDesc(src=W.data_ptr(), rows=W.shape[0], cols=W.shape[1], transpose=0, n_off=0, k_off=0, KS=KS,
                         dst=img.data_ptr(), fp8=1, scales=sc.data_ptr())
    table = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).clone().to(DEV)
    _lib.check(lib.hsimae_pack_matrix(table.data_ptr(), 1, W.numel(), stream()))
    torch.cuda.synchronize()
    return img, sc


def gemm(akind, epi, bm=0, **kw):
    p = _lib.GemmParams()
    for k, v in kw.items():
        setattr(p, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    p.prec = _lib.PREC_FP8
    _lib.check(_lib.load().hsimae_gemm_tiled(C.byref(p), akind, epi, bm, 0, stream()), "hsimae_gemm (fp8)")
    torch.cuda.synchronize()


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def padk(x, K128):
    out = torch.zeros(*x.shape[:-1], K128, device=x.device, dtype=torch.float32)
    out[..., :x.shape[-1]] = x.float()
    return out


@pytest.mark.parametrize("M,N,K,bm", [(300, 512, 512, 0), (77, 512, 1376, 64), (200, 512, 1536, 128), (130, 1376, 512, 0),
                                      (257, 256, 256, 0), (64, 512, 2752, 0)])
def test_mx_gemm_bf16_operand_against_emulation(M, N, K, bm):
    torch.manual_seed(0)
    # wide dynamic range along K: per-block scales really differ (a per-tensor scale would lose the small blocks)
    A = (torch.randn(M, K, device=DEV) * torch.exp2(torch.randint(-6, 4, (M, K // 32), device=DEV).float()).repeat_interleave(32, 1)
         ).to(torch.bfloat16)
    W = torch.randn(N, K, device=DEV) * 0.05 * torch.exp2(torch.randint(-3, 3, (N, 1), device=DEV).float())
    bias = torch.randn(N, device=DEV)
    N_img = rup(N, 16)
    img, sc = pack8(W, N_img, K)
    out = torch.full((M, N), float("nan"), device=DEV)
    gemm(_lib.A_BF16, _lib.E_F32, bm, A=A, lda=K, M=M, N=N_img, K=K, n_valid=N, W8=img, S8=sc, bias=bias, out=out, ldo=N)
    K128 = rup(K, 128)
    ref = (mx_e4m3(padk(A, K128)).double() @ mx_e4m3(padk(W, K128)).double().t() + bias.double()).float()
    assert torch.isfinite(out).all()
    assert rel(out, ref) < 1e-4        # accumulation order / the matrix core's internal alignment of the 128 products
    # and the quantisation itself is what fp8 costs: a few percent against the unquantised product
    exact = A.float() @ W.t() + bias
    assert 1e-3 < rel(out, exact) < 0.2


def test_mx_gemm_saturates_instead_of_nan():
    """Scaled values land in [256, 512): everything above 448 must clamp (v_cvt_pk_fp8_f32 alone turns > 464 into NaN)."""
    M, N, K = 64, 64, 128
    A = torch.full((M, K), 1.0, device=DEV)
    A[:, ::32] = 1.99                                   # block amax 1.99 -> scale 2^-8: 1.99 * 256 = 509 > 448
    A = A.to(torch.bfloat16)
    W = torch.eye(N, K, device=DEV)
    img, sc = pack8(W, N, K)
    out = torch.full((M, N), float("nan"), device=DEV)
    gemm(_lib.A_BF16, _lib.E_F32, 0, A=A, lda=K, M=M, N=N, K=K, n_valid=N, W8=img, S8=sc, out=out, ldo=N)
    assert torch.isfinite(out).all()
    ref = mx_e4m3(A.float()) @ mx_e4m3(W).t()
    assert rel(out, ref) < 1e-6 and abs(float(out[0, 0]) - 448 / 256) < 1e-6


def test_mx_gemm_layernorm_prologue_and_gate_epilogue():
    torch.manual_seed(1)
    M, K, h = 150, 512, 1368
    hp = rup(h, 32)
    x = torch.randn(M, K, device=DEV) * 2 + 0.3
    gam, bet = 1 + 0.1 * torch.randn(K, device=DEV), 0.1 * torch.randn(K, device=DEV)
    W1, W3 = torch.randn(h, K, device=DEV) * 0.04, torch.randn(h, K, device=DEV) * 0.04
    b1, b3 = torch.randn(h, device=DEV) * 0.1, torch.randn(h, device=DEV) * 0.1
    i1, s1 = pack8(W1, hp, K)
    i3, s3 = pack8(W3, hp, K)
    u = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    g = torch.full((M, hp), float("nan"), dtype=torch.bfloat16, device=DEV)
    h13 = torch.full((M, 2 * hp), float("nan"), dtype=torch.bfloat16, device=DEV)
    gemm(_lib.A_F32_LN, _lib.E_SWIGLU, 0, A=x, lda=K, M=M, N=hp, K=K, n_valid=h, W8=i1, S8=s1, W8b=i3, S8b=s3, bias=b1, bias2=b3,
         gamma=gam, beta=bet, u_out=u, ldu=K, out=g, ldo=hp, h13=h13, ldh=2 * hp, hoff=hp)
    un = torch.nn.functional.layer_norm(x, (K,), gam, bet, 1e-5)
    assert rel(u.float(), un) < 1e-2                                      # bf16 copy saved for the weight gradients
    uq = mx_e4m3(un).double()
    a1 = (uq @ mx_e4m3(W1).double().t() + b1.double()).float().to(torch.bfloat16).float()
    a3 = (uq @ mx_e4m3(W3).double().t() + b3.double()).float().to(torch.bfloat16).float()
    # LayerNorm in the kernel vs torch differs in the last fp32 bits, which can move an element across an e4m3 rounding
    # boundary: compare in RMS, and the pre-activations at bf16 resolution
    def rms(a, b):
        return float((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt())
    assert rms(h13[:, :h].float(), a1) < 3e-3 and rms(h13[:, hp:hp + h].float(), a3) < 3e-3
    assert rms(g[:, :h].float(), torch.nn.functional.silu(a1) * a3) < 1e-2
    assert float(g[:, h:].float().abs().max()) == 0.0                      # padded hidden columns are exact zeros


def huge(bands=192):
    with contextlib.redirect_stdout(io.StringIO()):
        return HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=bands, b_patch_size=8, embed_dim=512, depth=12, num_heads=32,
                      s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True)


def rms_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


@pytest.mark.parametrize("grid", [(6, 9), (9, 6), (18, 3)])
def test_huge_fp8_against_oracle(grid):
    cfg = O.OracleConfig(bands=192, embed_dim=512, num_heads=32)
    state = O.init_state(cfg, seed=0, std=0.02)
    N = 6
    g = torch.Generator().manual_seed(7)
    x = torch.rand(N, 1, 192, 9, 9, generator=g)
    n1, n2 = torch.rand(N, 24, generator=g), torch.rand(N, 9, generator=g)
    ref_loss, _, ref_mask, ref_grads = O.forward_backward(state, cfg, x, n1.numpy(), n2.numpy(), *grid)
    res = {}
    for prec in ("bf16", "fp8"):
        m = huge()
        m.load_state_dict(state)
        m = m.to(DEV).set_precision(prec)
        loss, pred, mask = m(x.to(DEV), 0.75, noise=(n1, n2), grid=grid)
        loss.backward()
        assert torch.equal(mask.cpu(), ref_mask), "masks must be bit-exact in every precision (the noise is fp32)"
        named = dict(m.named_parameters())
        errs = {k: rms_rel(named[k].grad, ref_grads[k]) for k in ref_grads if not k.endswith("attn.k.bias")}
        res[prec] = (abs(loss.item() - ref_loss.item()) / ref_loss.item(), max(errs.values()), max(errs, key=errs.get))
        sd = m.state_dict()
        assert all(v.dtype == torch.float32 for v in sd.values()) and len(sd) == 535       # compute copies never leak
    print(f"Huge@512 grid {grid}: loss rel bf16 {res['bf16'][0]:.2e} fp8 {res['fp8'][0]:.2e}; worst grad RMS-rel "
          f"bf16 {res['bf16'][1]:.3f} fp8 {res['fp8'][1]:.3f} ({res['fp8'][2]})")
    # (18, 3): the spatial stack attends over 3 tokens, its q / k weight gradients are nearly zero — RMS-relative error is
    # measured on rounding noise there (0.030 in bf16 on these inputs)
    assert res["bf16"][0] <= 1e-4 and res["bf16"][1] <= 3.5e-2
    assert res["fp8"][0] <= 2e-4, "stated fp8 loss tolerance (measured 1.8e-5 ... 3.2e-5; 2e-3 until round 6)"
    assert res["fp8"][1] <= 0.12


@pytest.mark.parametrize("grid", [(6, 9)])            # ((9, 6) measured the same: profiles/r05_g_mx_operand_oracle.txt)
def test_huge_fp8_error_is_the_mx_operand_rounding(grid):
    """VERDICT r04 weak spot 3 / "Next round" 6: the fp8 gates against the fp32 oracle are loose (6 % on gradient norms, 12-15 %
    RMS on elements) because that is what e4m3 operands cost — this test shows it.  The oracle re-run with the encoder linears'
    operands quantised as the kernels quantise them (`oracle.operands_mx8`: MX e4m3 blocks of 32 along the contraction axis in
    the forward AND the data-gradient products, bf16 weight-gradient operands, bf16 everywhere operands_bf16 rounds) lands on
    the HIP fp8 gradients several times closer than the fp32 oracle does: the gap to the reference is the operand format, what
    remains (quantisation decisions that flip on the last bf16 bit, accumulation order) is gated tightly."""
    cfg = O.OracleConfig(bands=192, embed_dim=512, num_heads=32)
    state = O.init_state(cfg, seed=0, std=0.02)
    N = 6
    g = torch.Generator().manual_seed(7)
    x = torch.rand(N, 1, 192, 9, 9, generator=g)
    n1, n2 = torch.rand(N, 24, generator=g), torch.rand(N, 9, generator=g)
    l32, _, _, g32 = O.forward_backward(state, cfg, x, n1.numpy(), n2.numpy(), *grid)
    with O.operands_mx8():
        l8, _, _, g8 = O.forward_backward(state, cfg, x, n1.numpy(), n2.numpy(), *grid)
    m = huge()
    m.load_state_dict(state)
    m = m.to(DEV).set_precision("fp8")
    loss, _, _ = m(x.to(DEV), 0.75, noise=(n1, n2), grid=grid)
    loss.backward()
    named = dict(m.named_parameters())
    keys = [k for k in g32 if not k.endswith("attn.k.bias")]
    e32 = {k: rms_rel(named[k].grad, g32[k]) for k in keys}           # HIP fp8 vs the reference arithmetic
    e8 = {k: rms_rel(named[k].grad, g8[k]) for k in keys}             # HIP fp8 vs the same operand format on the CPU
    gap = {k: rms_rel(g8[k], g32[k]) for k in keys}                   # what the format itself moves
    enc = [k for k in keys if k.startswith("blocks")]
    w32, w8, wg = max(e32.values()), max(e8.values()), max(gap.values())
    med = lambda d, ks: sorted(d[k] for k in ks)[len(ks) // 2]       # noqa: E731
    rl32, rl8 = abs(loss.item() - l32.item()) / l32.item(), abs(loss.item() - l8.item()) / l8.item()
    print(f"[mx operand oracle] Huge@512 grid {grid}: loss rel vs fp32 {rl32:.2e}, vs mx oracle {rl8:.2e}; encoder gradients RMS-rel "
          f"median / worst: HIP vs fp32 {med(e32, enc):.3f} / {w32:.3f} ({max(e32, key=e32.get)}), HIP vs mx oracle {med(e8, enc):.3f} / {w8:.3f} "
          f"({max(e8, key=e8.get)}), mx oracle vs fp32 {med(gap, enc):.3f} / {wg:.3f}")
    # measured (profiles/r05_g_mx_operand_oracle.txt): HIP vs fp32 median 0.046 / worst 0.074, mx oracle vs fp32 0.046 / 0.073,
    # HIP vs mx oracle median 0.002 / worst 0.022-0.036 (a decoder q bias: bf16 noise on a near-zero gradient), loss 1e-6
    assert wg >= 0.03                                  # the format alone moves gradients by several per cent ...
    assert med(e8, enc) <= 0.2 * med(e32, enc)         # ... and explains the HIP path's distance to the reference
    assert med(e8, enc) <= 6e-3 and w8 <= 0.05 and rl8 <= 2e-5      # what is left, gated at ~2x what was measured


def test_fp8_training_steps_track_bf16(monkeypatch):
    """Five AdamW steps in fp8 stay on the bf16 trajectory (loss within 2e-3 at every step), Base width (the path is generic).
    HSIMAE_FP8_UNFUSED=1: every encoder linear on the MX GEMMs, layer at a time — by default the d = 128 blocks keep their fused
    bf16 kernels under precision = fp8 (they are faster than the unfused fp8 form) and this test would compare bf16 with bf16."""
    from hsimae_amd import FusedAdamW
    monkeypatch.setenv("HSIMAE_FP8_UNFUSED", "1")
    losses = {}
    for prec in ("bf16", "fp8"):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=48, b_patch_size=8, embed_dim=128, depth=12, num_heads=8,
                       s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True)
        m = m.to(DEV).set_precision(prec)
        g = torch.Generator().manual_seed(3)
        x = torch.rand(32, 1, 48, 9, 9, generator=g).to(DEV)
        nz = [(torch.rand(32, 6, generator=g), torch.rand(32, 9, generator=g)) for _ in range(5)]
        opt = None
        ls = []
        for i in range(5):
            loss, _, _ = m(x, 0.75, noise=nz[i], grid=(2, 7))
            if opt is None:
                opt = FusedAdamW(m, lr=2e-3, weight_decay=5e-2, betas=(0.9, 0.95))
            opt.zero_grad()
            loss.backward()
            opt.step()
            ls.append(loss.item())
        losses[prec] = ls
    for a, b in zip(losses["bf16"], losses["fp8"]):
        assert abs(a - b) <= 2e-3 * abs(a), (losses["bf16"], losses["fp8"])
    assert losses["fp8"][-1] < losses["fp8"][0]
    assert losses["fp8"] != losses["bf16"]                # the fp8 leg really ran other arithmetic


def test_fp8_default_schedule_keeps_fused_kernels_at_base_width():
    """precision = fp8 at d = 128: every linear sits inside a fused bf16 kernel, so the default schedule computes exactly what
    bf16 computes (deterministic mode: bit-identical loss) — fp8 is never slower than bf16 at a width it is allowed on."""
    out = {}
    for prec in ("bf16", "fp8"):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=48, b_patch_size=8, embed_dim=128, depth=12, num_heads=8,
                       s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True)
        m = m.to(DEV).set_precision(prec)
        m.deterministic = True
        g = torch.Generator().manual_seed(3)
        x = torch.rand(16, 1, 48, 9, 9, generator=g).to(DEV)
        nz = (torch.rand(16, 6, generator=g), torch.rand(16, 9, generator=g))
        loss, _, _ = m(x, 0.75, noise=nz, grid=(2, 7))
        loss.backward()
        out[prec] = (loss.item(), dict(m.named_parameters())["blocks_1.3.mlp.w1.weight"].grad.clone())
    assert out["bf16"][0] == out["fp8"][0] and torch.equal(out["bf16"][1], out["fp8"][1])
