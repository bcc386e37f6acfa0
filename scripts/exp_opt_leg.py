"""Per-iteration wall time of bench.py's optimizer leg (FusedAdamW.step + packed-weight refresh) at Huge, to find the step that
occasionally takes ~90 ms.   python scripts/exp_opt_leg.py [bf16|fp8]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hsimae_amd import HSIMAE, FusedAdamW  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    bands, D, heads, N = bench.MODELS["huge"]
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=bands, b_patch_size=8, embed_dim=D, depth=12, num_heads=heads, s_depth=9,
               decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True).cuda()
    if prec == "fp8":
        m.set_precision("fp8")
    x = torch.rand(N, 1, bands, 9, 9, device="cuda")
    for rep in range(3):
        print(bench.optimizer_step_ms(m, x), flush=True)
    opt = FusedAdamW(m, lr=1e-9)
    stream = torch.cuda.current_stream().cuda_stream
    m.zero_grad(set_to_none=True)
    loss, _, _ = m(x, 0.75)
    loss.backward()
    ts = []
    for i in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.step()
        t1 = time.perf_counter()
        m._ensure_packed(stream)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    for i, (a, b, c) in enumerate(ts):
        print(f"{i:2d} step() host {a * 1e3:7.3f} ms  pack host {b * 1e3:7.3f} ms  drain {c * 1e3:7.3f} ms")


if __name__ == "__main__":
    main()
