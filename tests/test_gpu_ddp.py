"""Data-parallel gradient equality on a real MI355X (tier 4): two ranks, each with half of a global batch, must end
with the gradients a single process computes on the whole batch.

The GPU box has ONE GPU and RCCL refuses two ranks on one device, so the two ranks share cuda:0 and talk over **gloo**
(device tensors, staged through the host).  Everything else is the production path: `enable_data_parallel()` (broadcast of
rank 0's weights), the bucket callback fired from inside `hsimae_backward`, the reducer's launch stream made to wait on the
library's events (side-stream ranges included), bucket merging of the interleaved axis-stack ranges, 1/world folded into
dLoss/dpred.  The RCCL transport itself runs in `bench.py --force-ddp` (one rank) and in the driver's multi-GPU runs.
"""
import contextlib
import io
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(seed):
    from hsimae_amd import HSIMAE
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        return HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=48, b_patch_size=8, embed_dim=128, depth=12, num_heads=8,
                      s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True)


def _inputs(N):
    g = torch.Generator().manual_seed(123)
    return torch.rand(N, 1, 48, 9, 9, generator=g), torch.rand(N, 6, generator=g), torch.rand(N, 9, generator=g)


def _worker(rank, world, port, q, backend="gloo", split_api=False):
    """backend "gloo": both ranks share cuda:0 (the 1-GPU box); "nccl": one GPU per rank over RCCL (needs >= 2 GPUs).
    split_api: the step is composed from forward_encoder / forward_decoder / forward_loss (three autograd nodes, two
    reducer-driven backward passes) instead of the one-node forward()."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        dev = torch.device("cuda", rank if backend == "nccl" else 0)
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        m = _model(seed=100 + rank).to(dev)              # different weights per rank: the broadcast must fix that
        m.enable_data_parallel(bucket_bytes=1 << 20)
        m.deterministic = True                            # fixed-point reduction: the comparison is not blurred by atomics
        N = 32
        x, n1, n2 = _inputs(N)
        per = N // world
        sl = slice(rank * per, (rank + 1) * per)
        losses = []
        for step in range(2):                             # two steps: the second reuses arena, events and launch stream
            m.zero_grad(set_to_none=True)
            if split_api:
                xs = x[sl].to(dev)
                latent, mask, ids_restore, _ = m.forward_encoder(xs, 0.75, noise=(n1[sl], n2[sl]), grid=(2, 7))
                loss = m.forward_loss(xs, m.forward_decoder(latent, ids_restore), mask)
            else:
                loss, _, _ = m(x[sl].to(dev), 0.75, noise=(n1[sl], n2[sl]), grid=(2, 7))
            loss.backward()
            torch.cuda.synchronize()
            losses.append(loss.item())
        # numpy (pickled by value): torch tensors travel through the queue as shared-memory handles that die with the worker
        grads = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
        launched = list(m._reducer.launched)
        sd0 = m.state_dict()["blocks.0.mlp.w1.weight"].cpu().numpy()
        q.put((rank, "ok", losses, grads if rank == 0 else None, launched, sd0))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc() + repr(e), None, None, None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _two_gpus():
    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


@pytest.mark.parametrize("backend,split_api", [
    ("gloo", False),
    ("gloo", True),          # the split API's loss gradient must carry 1/world too
    pytest.param("nccl", False, marks=pytest.mark.skipif(not _two_gpus(), reason="RCCL needs one GPU per rank (>= 2 GPUs)")),
    pytest.param("nccl", True, marks=pytest.mark.skipif(not _two_gpus(), reason="RCCL needs one GPU per rank (>= 2 GPUs)")),
])
def test_two_ranks_equal_one_process_on_the_global_batch(backend, split_api):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend, split_api)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    assert [r[1] for r in res] == ["ok", "ok"], [r[1] for r in res]
    # rank 0's weights reached rank 1
    assert (res[0][5] == res[1][5]).all()
    # the collectives cover the trainable part of the flat buffer in a few buckets, identically on both ranks
    # (split API: `launched` is what the last of its two backward passes — the encoder's — issued)
    assert res[0][4] == res[1][4] and (2 if split_api else 3) <= len(res[0][4]) <= 40
    # single process, whole batch, rank 0's weights
    m = _model(seed=100).cuda()
    m.deterministic = True
    x, n1, n2 = _inputs(32)
    loss, _, _ = m(x.cuda(), 0.75, noise=(n1, n2), grid=(2, 7))
    loss.backward()
    torch.cuda.synchronize()
    # equal sum(mask) per rank => mean of the rank losses == the global masked mean
    mean_rank_loss = 0.5 * (res[0][2][1] + res[1][2][1])
    assert abs(mean_rank_loss - loss.item()) <= 2e-6 * abs(loss.item())
    named = dict(m.named_parameters())
    worst = ("", 0.0)
    for k, g in res[0][3].items():
        ref = named[k].grad.detach().cpu()
        e = float((torch.from_numpy(g) - ref).abs().max() / ref.abs().max().clamp_min(1e-20))
        if not k.endswith("attn.k.bias") and e > worst[1]:
            worst = (k, e)
    # same arithmetic per sample, different partition of the row sums (fp32 partials inside each rank's kernels)
    assert worst[1] < 2e-3, worst


def _cubes():
    import numpy as np
    from oracle import loader_oracle as LO
    rng = np.random.default_rng(5)
    scenes = [rng.random((16, 17, 32)).astype(np.float32), rng.random((13, 15, 32)).astype(np.float32)]
    cut = []
    for num, sc in enumerate(scenes):
        cut += LO.split_info(sc.shape, (9, 9, 32), (3, 3, 1), num, 1, 0)
    return [scenes, np.array(cut[:24], dtype=np.int16)]      # 24 cubes = 3 global batches of 8: nothing ragged to drop


_KW = dict(img_size=9, bands=32, mask_ratio=0.5, lr=5e-3, wd=5e-2, bs=8, depth=3, dim=32, s_depth=2, dec_dim=32, dec_depth=2,
           log=lambda *_: None)


def _train_worker(rank, world, port, out_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSIMAE_DP_BACKEND="gloo", HSIMAE_DETERMINISTIC="1")
    try:
        import hsimae_amd
        from hsimae_amd.pretrain import seed_everything
        seed_everything(0)
        with contextlib.redirect_stdout(io.StringIO()):
            model, losses = hsimae_amd.mask_pretraining(_cubes(), out_dir, "m.pkl", epochs=2, device="cuda:0", **_KW)
        w = model.state_dict()["blocks.0.mlp.w2.weight"].cpu().numpy()
        q.put((rank, "ok", losses, w, os.path.exists(os.path.join(out_dir, "m.pkl"))))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc() + repr(e), None, None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_mask_pretraining_two_ranks_tracks_the_single_process_run(tmp_path):
    """The training entry point under a 2-rank launch (global batch 8 = 2 x 4): same permutation, flips, grid and masking
    noise as the single-process run at batch 8, gradients averaged by the reducer -> the same loss trajectory and weights."""
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    out = str(tmp_path / "dp")
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, out, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    assert [r[1] for r in res] == ["ok", "ok"], [r[1] for r in res]
    assert res[0][4] and np.array_equal(res[0][3], res[1][3])          # rank 0 wrote the files; the ranks hold the same weights
    assert np.allclose(res[0][2], res[1][2])                            # the logged epoch loss is the all-reduced mean
    # single process, same global batch
    import hsimae_amd
    from hsimae_amd.pretrain import seed_everything
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    os.environ["HSIMAE_DETERMINISTIC"] = "1"
    try:
        seed_everything(0)
        with contextlib.redirect_stdout(io.StringIO()):
            model, losses = hsimae_amd.mask_pretraining(_cubes(), str(tmp_path / "sp"), "m.pkl", epochs=2, device="cuda:0", **_KW)
    finally:
        os.environ.pop("HSIMAE_DETERMINISTIC", None)
    assert np.allclose(res[0][2], losses, rtol=2e-3), (res[0][2], losses)
    w = model.state_dict()["blocks.0.mlp.w2.weight"].cpu().numpy()
    assert np.abs(res[0][3] - w).max() <= 2e-2 * np.abs(w).max()


@pytest.mark.skipif(not _two_gpus(), reason="needs >= 2 GPUs")
def test_bench_two_gpus_prints_one_line_with_comm():
    """`python bench.py --gpus 2` starts its own two ranks over RCCL and rank 0 prints ONE JSON line (n_gpus 2, weak scaling,
    the `comm` object that explains the all-reduce cost)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["comm"]["buckets"] >= 1 and rec["comm"]["bytes"] > 0
