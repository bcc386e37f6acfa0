"""Data-parallel gradient reducer on CPU: world_size 2 over gloo (the N>1 path without GPUs).

Covers: rank discovery, parameter broadcast, bucketed all-reduce driven by the same (off, len) range stream the
backward schedule emits (back to front), sum == mean because each rank pre-scales by 1/world, and that every
element of the flat gradient buffer is reduced exactly once."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hsimae_amd.parallel import GradReducer
        red = GradReducer(bucket_bytes=4 * 300)
        assert red.world_size == world and red.rank == rank
        # parameter broadcast from rank 0
        p = torch.full((1000,), float(rank + 1))
        red.broadcast(p)
        assert torch.equal(p, torch.ones(1000))
        # gradients: rank r holds (r+1) * base, pre-scaled by 1/world  => reduced value = base * mean(r+1)
        sizes = [64, 128, 128, 500, 128, 40, 12]
        offs = [sum(sizes[:i]) for i in range(len(sizes))]
        total = sum(sizes)
        base = torch.arange(total, dtype=torch.float32)
        flat = base * (rank + 1) / world
        # callback-driven path (what hsimae_backward does), ranges arrive back to front
        cb = red.make_callback(flat)
        for st, i in enumerate(reversed(range(len(sizes)))):
            red._on_range(st, offs[i], sizes[i], None)
        red.finish()
        want = base * sum(r + 1 for r in range(world)) / world
        assert torch.allclose(flat, want), (flat - want).abs().max()
        # plan-driven path
        flat2 = base * (rank + 1) / world
        red.reduce_ranges(flat2, [(offs[i], sizes[i]) for i in reversed(range(len(sizes)))])
        assert torch.allclose(flat2, want)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
