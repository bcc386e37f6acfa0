"""Time of the packed-weight refresh (hsimae_pack_params) and of FusedAdamW.step alone, per model / precision.
    python scripts/exp_pack_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hsimae_amd import HSIMAE, FusedAdamW  # noqa: E402


def main():
    for model, prec in (("base", "bf16"), ("large", "bf16"), ("huge", "bf16"), ("huge", "fp8")):
        bands, D, heads, _ = bench.MODELS[model]
        m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=bands, b_patch_size=8, embed_dim=D, depth=12, num_heads=heads, s_depth=9,
                   decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True).cuda()
        if prec == "fp8":
            m.set_precision("fp8")
        x = torch.rand(64, 1, bands, 9, 9, device="cuda")
        loss, _, _ = m(x, 0.75)
        loss.backward()
        opt = FusedAdamW(m, lr=1e-9)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            opt.step(); m._ensure_packed(stream)
        torch.cuda.synchronize()
        t = {}
        for name, fn in (("adamw", lambda: opt.step()), ("pack", lambda: (setattr(m, "_packed_version", -1), m._ensure_packed(stream)))):
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t[name] = (time.perf_counter() - t0) / 10 * 1e3
        print(f"{model:6s} {prec}: FusedAdamW.step {t['adamw']:.3f} ms   hsimae_pack_params {t['pack']:.3f} ms", flush=True)
        del m, opt


if __name__ == "__main__":
    main()
