"""Soak (GPU box): 150 training steps per mode (bf16, fp8, deterministic, DDP-path single rank is covered by bench) with
alternating batch sizes and a monitoring forward every 10 steps; checks finite decreasing loss and a flat memory footprint
(the workspace pool must not grow).   python scripts/soak_r02.py"""
import contextlib, io, os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsimae_amd import HSIMAE, FusedAdamW

def run(prec, det):
    torch.manual_seed(0); random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, embed_dim=128, depth=12, num_heads=8, s_depth=9,
                   decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True).cuda()
    m.set_precision(prec); m.deterministic = det
    x = torch.rand(512, 1, 96, 9, 9, device="cuda")
    opt, first, last, peak0 = None, None, None, None
    for i in range(150):
        n = 512 if i % 7 else 300                       # a ragged batch now and then: another arena
        loss, _, _ = m(x[:n], 0.75)
        if opt is None:
            opt = FusedAdamW(m, lr=1e-3, weight_decay=5e-2, betas=(0.9, 0.95))
        opt.zero_grad(); loss.backward(); opt.step()
        if i % 10 == 0:
            with torch.no_grad():
                m(x[:64], 0.75)
        v = loss.item()
        assert v == v and v < 10, (prec, det, i, v)
        first = v if first is None else first
        last = v
        if i == 30:
            torch.cuda.synchronize(); peak0 = torch.cuda.memory_allocated()
    torch.cuda.synchronize()
    grow = torch.cuda.memory_allocated() - peak0
    print(f"{prec:5s} det={det}: loss {first:.4f} -> {last:.4f}, allocated after step 30 vs end: {grow / 1e6:+.1f} MB", flush=True)
    assert last < first and abs(grow) < 64e6

for prec, det in (("bf16", False), ("fp8", False), ("bf16", True)):
    run(prec, det)
print("SOAK_OK")
