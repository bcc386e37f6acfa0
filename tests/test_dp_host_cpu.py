"""Host side of the data-parallel path and of the loader's RNG bookkeeping, on CPU (no GPU, no kernels):

  * `bench.py --gpus 2` starts its own two ranks (gloo dry run of the launcher, barrier / MAX-over-ranks protocol);
  * `DeviceLoader` draws the default generator exactly where torch's DataLoader does, also when two loaders are
    created back to back and advanced in turn (Model_Finetuning.py:142-149);
  * the per-rank shard of `DeviceLoader(rank, world)` under a world-2 gloo launch: union over ranks == the
    single-process batch sequence, identical python-random stream on every rank;
  * `GradReducer`: ranges of the two axis stacks arriving interleaved are merged into buckets, every element reduced
    once, and an exception inside the ctypes callback surfaces from finish().
"""
import json
import os
import random
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class FakeCubes:
    """Stands in for HSIdataset4PT where only the index / RNG bookkeeping is under test."""
    train = True

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def batch(self, idx):
        from hsimae_amd.data import draw_flips
        return list(idx), draw_flips(len(idx), True).tolist()

    def gather(self, idx, flips):
        return list(idx), np.asarray(flips).tolist()


def test_bench_launches_its_own_ranks():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    # the slowest rank sleeps 2 x 2 ms per step: the reported time is the MAX over ranks
    assert out["ms_per_step"] >= 3.9


def test_bench_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode == 2


def test_loader_rng_matches_dataloader_when_two_iterators_interleave():
    from torch.utils.data import DataLoader
    from hsimae_amd.data import DeviceLoader
    n1, n2, bs = 37, 23, 8

    def torch_side():
        torch.manual_seed(11)
        a = iter(DataLoader(list(range(n1)), batch_size=bs, shuffle=True))
        torch.manual_seed(12)
        b = iter(DataLoader(list(range(n2)), batch_size=bs, shuffle=True))
        out = []
        for _ in range(2):
            out.append(next(a).tolist())
            out.append(next(b).tolist())
        out.append(int(torch.empty((), dtype=torch.int64).random_().item()))      # where the default generator stands
        return out

    def our_side():
        torch.manual_seed(11)
        a = iter(DeviceLoader(FakeCubes(n1), batch_size=bs, shuffle=True))
        torch.manual_seed(12)
        b = iter(DeviceLoader(FakeCubes(n2), batch_size=bs, shuffle=True))
        out = []
        for _ in range(2):
            out.append(next(a)[0])
            out.append(next(b)[0])
        out.append(int(torch.empty((), dtype=torch.int64).random_().item()))
        return out

    assert torch_side() == our_side()
    # single loader, whole epoch incl. the ragged last batch and a non-shuffled loader's base-seed draw
    torch.manual_seed(5)
    ref = [b.tolist() for b in DataLoader(list(range(n1)), batch_size=bs, shuffle=True)]
    ref_ns = [b.tolist() for b in DataLoader(list(range(n2)), batch_size=bs, shuffle=False)]
    ref_pos = int(torch.empty((), dtype=torch.int64).random_().item())
    torch.manual_seed(5)
    got = [b[0] for b in DeviceLoader(FakeCubes(n1), batch_size=bs, shuffle=True)]
    got_ns = [b[0] for b in DeviceLoader(FakeCubes(n2), batch_size=bs, shuffle=False)]
    assert got == ref and got_ns == ref_ns and int(torch.empty((), dtype=torch.int64).random_().item()) == ref_pos


def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        from hsimae_amd.data import DeviceLoader
        from hsimae_amd.pretrain import dp_context, seed_everything
        r, w, dev = dp_context(torch.device("cpu"))          # creates the gloo group from the launcher's environment
        assert (r, w) == (rank, world) and dist.is_initialized()
        n, gbs = 53, 12
        seed_everything(42)
        ld = DeviceLoader(FakeCubes(n), batch_size=gbs // w, shuffle=True, rank=r, world=w)
        mine = [b for b in ld]
        py_pos = random.random()                              # python-random stream position after the epoch
        # single-process reference at the global batch, same seeds
        seed_everything(42)
        full = [b for b in DeviceLoader(FakeCubes(n), batch_size=gbs, shuffle=True)]
        assert py_pos == random.random(), "python-random stream differs from the single-process run"
        assert len(mine) == len(ld) and len(ld) in (len(full), len(full) - 1)      # a ragged tail of < world cubes is dropped
        gathered = [None] * w
        dist.all_gather_object(gathered, mine)
        for step, (fi, ff) in enumerate(full):
            per = len(fi) // w
            cat_i = sum((gathered[k][step][0] for k in range(w)), [])
            cat_f = sum((gathered[k][step][1] for k in range(w)), [])
            assert cat_i == fi[:per * w] and cat_f == ff[:per * w]        # ragged tail: < world cubes dropped
            assert all(len(gathered[k][step][0]) == per for k in range(w))   # equal shares => equal sum(mask) per rank
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_rank_shards_cover_the_global_batch_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _reducer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hsimae_amd.parallel import GradReducer
        # layout like the flat gradient buffer: [head | b1_0 b1_1 b1_2 | b2_0 b2_1 b2_2 | tail]; the backward reports
        # tail, then b2_2, b1_2, b2_1, b1_1, b2_0, b1_0 (the two stacks interleaved), then head
        sizes = dict(head=40, b1=[100, 100, 100], b2=[100, 100, 100], tail=70)
        off, cur = {}, 0
        off["head"] = cur; cur += sizes["head"]
        off["b1"] = []
        for s in sizes["b1"]:
            off["b1"].append(cur); cur += s
        off["b2"] = []
        for s in sizes["b2"]:
            off["b2"].append(cur); cur += s
        off["tail"] = cur; cur += sizes["tail"]
        total = cur
        base = torch.arange(total, dtype=torch.float32)
        flat = base * (rank + 1) / world
        red = GradReducer(bucket_bytes=4 * 150)
        red.make_callback(flat)
        st = 0
        red._on_range(st, off["tail"], sizes["tail"], None)
        for i in (2, 1, 0):
            st += 1; red._on_range(st, off["b2"][i], 100, None)
            st += 1; red._on_range(st, off["b1"][i], 100, None)
        red._on_range(st + 1, off["head"], sizes["head"], None)
        red.finish()
        want = base * sum(r + 1 for r in range(world)) / world
        assert torch.allclose(flat, want), (flat - want).abs().max()
        cover = sorted(red.launched)
        assert cover[0][0] == 0 and cover[-1][1] == total and all(a[1] == b[0] for a, b in zip(cover, cover[1:])), cover
        assert len(cover) < 8                                 # merged into buckets, not one collective per range
        # an exception raised inside the callback is not lost (ctypes would print and swallow it)
        red.make_callback(flat)
        red._flat = None                                      # slicing None raises inside _launch
        red._on_range(0, 0, total, None)
        try:
            red.finish()
            q.put((rank, "finish() did not raise"))
            return
        except RuntimeError as e:
            assert "all-reduce failed" in str(e)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


def test_reducer_merges_interleaved_stack_ranges_and_reports_errors_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
