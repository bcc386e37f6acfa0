"""Where the ~4 % single-rank slowdown of the torch.distributed path comes from (GPU box, one rank):
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 scripts/ddp_tax.py"""
import contextlib, io, os, random, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsimae_amd import HSIMAE

def bench(m, x, tag):
    def run(n):
        for _ in range(n):
            m.zero_grad(set_to_none=True)
            loss, _, _ = m(x, 0.75); loss.backward()
    random.seed(0); run(4); torch.cuda.synchronize(); t = time.perf_counter(); run(15); torch.cuda.synchronize()
    print(f"{tag:40s} {(time.perf_counter() - t) / 15 * 1e3:7.3f} ms", flush=True)

torch.cuda.set_device(0)
with contextlib.redirect_stdout(io.StringIO()):
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, embed_dim=128, depth=12, num_heads=8, s_depth=9,
               decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True).cuda()
x = torch.rand(4096, 1, 96, 9, 9, device="cuda")
bench(m, x, "before init_process_group")
mode = os.environ.get("TAX_MODE", "lazy")
if mode == "eager":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("nccl")
bench(m, x, f"after init_process_group ({mode})")
t = torch.ones(8, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
bench(m, x, "after the first all_reduce")
m.enable_data_parallel()
bench(m, x, "with the reducer attached")
dist.destroy_process_group()
bench(m, x, "after destroy_process_group")
