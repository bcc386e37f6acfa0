"""Soak run on the GPU box: 300 optimizer steps of HSIMAE-Base at batch 4096 (FusedAdamW + cosine schedule) on structured
synthetic cubes; prints the loss trajectory, a finiteness check and the wall time per iteration including the optimizer."""
import sys, time, torch, random, contextlib, io
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsimae_amd import HSIMAE, FusedAdamW, CosineLRScheduler
torch.manual_seed(0); random.seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, embed_dim=128, depth=12, num_heads=8, s_depth=9,
               decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True).cuda()
# structured synthetic cubes: smooth spectra + spatial pattern so there is something to learn
N = 4096
g = torch.Generator(device='cuda').manual_seed(1)
base = torch.rand(N, 1, 1, 1, 1, device='cuda', generator=g)
spec = torch.sin(torch.linspace(0, 6.28, 96, device='cuda')).view(1, 1, 96, 1, 1) * torch.rand(N, 1, 1, 1, 1, device='cuda', generator=g)
spat = torch.rand(N, 1, 1, 9, 9, device='cuda', generator=g) * 0.3
x = (0.5 * base + 0.25 * spec + spat + 0.25).clamp(0, 1).contiguous()
opt = FusedAdamW(m, lr=2e-3, weight_decay=5e-2, betas=(0.9, 0.95))
iters = 300
sch = CosineLRScheduler(opt, t_initial=iters, lr_min=1e-6, warmup_t=15)
t0 = time.perf_counter(); losses = []
for it in range(iters):
    loss, _, _ = m(x, mask_ratio=0.75)
    opt.zero_grad(); loss.backward(); opt.step(); sch.step(it)
    if it % 30 == 0 or it == iters - 1: losses.append(round(loss.item(), 4))
torch.cuda.synchronize()
print("losses", losses, "finite", all(l == l for l in losses), f"{(time.perf_counter()-t0)/iters*1e3:.2f} ms/iter incl. optimizer")
