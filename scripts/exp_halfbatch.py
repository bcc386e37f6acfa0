"""Scheduling experiment (round 4): does the C2 step get shorter when the batch runs as two half-batch pipelines on two
streams?  The encoder's backward kernels are HBM-bound (4.1-4.9 TB/s, MFMA 10-17 % busy), the decoder's are issue / latency
bound (1.6-2.1 TB/s): kernels of the two kinds running side by side would use complementary resources.  No kernel changes:
two model instances (own flat buffers, own workspace) with the same weights, each on its own stream.

    python scripts/exp_halfbatch.py [--batch 4096] [--steps 20]
prints ms per step for: one model on the whole batch; two halves back to back on one stream; two halves on two streams
(forwards enqueued first, then the backwards); the same with quarter batches on four streams.
"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hsimae_amd import HSIMAE  # noqa: E402


def make(dev):
    torch.manual_seed(0)
    return HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, embed_dim=128, depth=12, num_heads=8,
                  s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True,
                  trunc_init=True).to(dev)


def timed(fn, steps, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    N = args.batch
    imgs = torch.rand(N, 1, 96, 9, 9, device=dev)
    grid = (3, 9)
    models = [make(dev) for _ in range(4)]
    for m in models:
        m.want_recons = True
    streams = [torch.cuda.Stream(dev) for _ in range(4)]

    def whole():
        m = models[0]
        m.zero_grad(set_to_none=True)
        loss, _, _ = m(imgs, 0.75, grid=grid)
        loss.backward()

    def parts(k, concurrent):
        n = N // k
        cur = torch.cuda.current_stream(dev)
        losses = []
        for i in range(k):
            s = streams[i] if concurrent else cur
            if concurrent:
                s.wait_stream(cur)
            with torch.cuda.stream(s):
                models[i].zero_grad(set_to_none=True)
                losses.append(models[i](imgs[i * n:(i + 1) * n], 0.75, grid=grid)[0])
        for i in range(k):
            s = streams[i] if concurrent else cur
            with torch.cuda.stream(s):
                (losses[i] * (1.0 / k)).backward()
        if concurrent:
            for i in range(k):
                cur.wait_stream(streams[i])

    random.seed(0)
    print(f"batch {N}, HSIMAE_TWO_STREAMS={os.environ.get('HSIMAE_TWO_STREAMS', '(default)')}")
    print(f"whole batch, one model              {timed(whole, args.steps):8.3f} ms")
    print(f"2 halves back to back, one stream   {timed(lambda: parts(2, False), args.steps):8.3f} ms")
    print(f"2 halves on two streams             {timed(lambda: parts(2, True), args.steps):8.3f} ms")
    print(f"4 quarters on four streams          {timed(lambda: parts(4, True), args.steps):8.3f} ms")
    print(f"whole batch again                   {timed(whole, args.steps):8.3f} ms")


if __name__ == "__main__":
    main()
