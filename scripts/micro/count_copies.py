import torch, sys, contextlib, io
sys.path.insert(0, '/root/repo')
from hsimae_amd import HSIMAE
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, embed_dim=128, depth=12, num_heads=8, s_depth=9,
               decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True).cuda()
x = torch.rand(512, 1, 96, 9, 9, device='cuda')
for _ in range(2):
    m.zero_grad(set_to_none=True); loss, _, _ = m(x, 0.75); loss.backward()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    m.zero_grad(set_to_none=True); loss, _, _ = m(x, 0.75); loss.backward(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if 'copy' in e.key.lower() or 'memcpy' in e.key.lower() or 'Memcpy' in e.key]
for e in sorted(prof.key_averages(), key=lambda e: -e.count)[:25]:
    print(f"{e.count:6d}  {e.key[:90]}")
