#!/usr/bin/env python3
"""Relative loss error of the HIP path vs the fp32 CPU oracle over seeds and weight scales (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hsimae_amd import HSIMAE
from oracle import hsimae_oracle as O

cfg = O.OracleConfig(bands=48)
for std in (0.02, 0.08):
    errs = []
    for seed in range(1, 7):
        state = O.init_state(cfg, seed=seed, std=std)
        g = torch.Generator().manual_seed(100 + seed)
        N = 32
        x = torch.rand(N, 1, 48, 9, 9, generator=g)
        n1, n2 = torch.rand(N, 6, generator=g), torch.rand(N, 9, generator=g)
        ref, _, _ = O.forward(state, cfg, x, n1.numpy(), n2.numpy(), 2, 7)
        m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=48, b_patch_size=8, embed_dim=128, depth=12, num_heads=8,
                   s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True)
        m.load_state_dict(state); m = m.to("cuda:0")
        with torch.no_grad():
            loss = m(x.cuda(), 0.75, noise=(n1, n2), grid=(2, 7))[0]
        errs.append((loss.item() - ref.item()) / ref.item())
    print(f"std={std}: rel loss err per seed:", " ".join(f"{e:+.2e}" for e in errs))
