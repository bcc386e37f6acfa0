"""Rank-count-dependent behaviour of the data-parallel path on CPU (no multi-GPU node has run it yet):

  * the gradient reducer at world 4 and 8 over gloo, driven by the REAL range stream of the C2 model — the order in which
    `hsimae_backward` (csrc/api.hip: decoder_backward, hsimae_backward, encoder_backward) reports parameter ranges, rebuilt here
    from `hsimae_param_layout` and the module tree, with the two axis stacks interleaved as the two-stream schedule reports
    them — into the default 4 MiB buckets: every trainable element is all-reduced exactly once, frozen tables never, the
    bucket count and byte volume are what the 1-rank RCCL run on the GPU measured (5 buckets, 18.5 MB), identically on every rank;
  * `bench.py --gpus 4 / 8 --dry-run`: the launcher, the barrier / MAX-over-ranks protocol and the one-JSON-line contract;
  * `DeviceLoader` shards at world 4 / 8 with a ragged last batch: the union over ranks is the single-process batch sequence
    minus the < world leftover cubes.
"""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

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


def c2_range_stream():
    """(off, len) in the order hsimae_backward emits them for HSIMAE-Base / 96 bands, plus the layout.  Mirrors api.hip:
    [decoder_norm .. end), decoder_blocks 7..0, [norm .. decoder_embed.bias], blocks 2..0, then for i = 8..0: blocks_2.i, blocks_1.i
    (the forked schedule reports each block when its kernels are enqueued), finally [0, patch_embed.proj.bias]."""
    import contextlib
    import io
    from hsimae_amd import HSIMAE, _lib
    with contextlib.redirect_stdout(io.StringIO()):
        m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, embed_dim=128, depth=12, num_heads=8, s_depth=9,
                   decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True)
    names = [n for n, _ in m.named_parameters()]
    params = [p for _, p in m.named_parameters()]
    n = len(names)
    offs, sizes = (C.c_int64 * n)(), (C.c_int64 * n)()
    assert _lib.load().hsimae_param_layout(C.byref(m._config()), offs, sizes, n) == n
    off = {k: (offs[i], offs[i] + sizes[i]) for i, k in enumerate(names)}
    total = offs[n - 1] + sizes[n - 1]

    def span(prefix):
        ks = [k for k in names if k.startswith(prefix)]
        return min(off[k][0] for k in ks), max(off[k][1] for k in ks)

    stream = [(off["decoder_norm.weight"][0], total)]
    stream += [span(f"decoder_blocks.{i}.") for i in range(7, -1, -1)]
    stream += [(off["norm.weight"][0], off["decoder_embed.bias"][1])]
    stream += [span(f"blocks.{i}.") for i in range(2, -1, -1)]
    for i in range(8, -1, -1):
        stream += [span(f"blocks_2.{i}."), span(f"blocks_1.{i}.")]
    stream += [(0, off["patch_embed.proj.bias"][1])]
    trainable = torch.zeros(total, dtype=torch.bool)
    for k, p in zip(names, params):
        if p.requires_grad and k != "mask_token":
            trainable[off[k][0]:off[k][1]] = True
    return [(a, b - a) for a, b in stream], total, trainable


def _reducer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hsimae_amd.parallel import GradReducer
        stream, total, trainable = c2_range_stream()
        red = GradReducer()                                   # default 4 MiB buckets, as enable_data_parallel() makes it
        base = (torch.arange(total, dtype=torch.float32) % 977) + 1.0
        flat = base * (rank + 1) / world                      # rank r holds (r + 1) * base, pre-scaled by 1 / world
        red.make_callback(flat)
        for st, (o, ln) in enumerate(stream):
            red._on_range(st, o, ln, None)
        red.finish()
        mean = sum(r + 1 for r in range(world)) / world
        covered = torch.zeros(total, dtype=torch.int32)
        for lo, hi in red.launched:
            covered[lo:hi] += 1
        assert int(covered.max()) == 1, "an element was reduced twice"
        assert bool((covered[trainable] == 1).all()), "a trainable element was not reduced"
        want = torch.where(covered.bool(), base * mean, base * (rank + 1) / world)
        assert torch.allclose(flat, want, rtol=1e-6), float((flat - want).abs().max())
        nbytes = int(sum(hi - lo for lo, hi in red.launched)) * 4
        q.put((rank, "ok", len(red.launched), nbytes, tuple(red.launched)))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e), 0, 0, ()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_reducer_with_the_c2_bucket_plan_at_world_4_and_8(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == ["ok"] * world, res
    assert len({r[4] for r in res}) == 1, "ranks disagree on the collectives they issue"
    nb, nbytes = res[0][2], res[0][3]
    # what bench.py --force-ddp measured on the GPU with one rank (profiles/r03_i_bench_base_ddp_path_1rank.json): 5 buckets, 18.5 MB
    assert nb == 5 and abs(nbytes - 18.5e6) < 0.2e6, (nb, nbytes)


@pytest.mark.parametrize("world", [4, 8])
def test_bench_dry_run_at_world_4_and_8(world):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-run", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 3 and out["scaling"] == "weak" and out["config"]["parallelism"] == f"dp{world}"
    assert out["ms_per_step"] >= 2.0 * world - 0.1        # the slowest rank sleeps 2 ms x world per step: MAX over ranks


@pytest.mark.parametrize("world", [4, 8])
def test_loader_shards_cover_the_global_batches_with_a_ragged_tail(world):
    """`DeviceLoader(rank, world)`: every rank draws the reference's permutation, rank r takes the r-th equal slice of each global
    batch; a ragged last batch drops its < world leftover cubes.  No process group needed: the shard is a pure
    function of (rank, world) and the shared RNG stream."""
    import random
    import numpy as np
    from hsimae_amd.data import DeviceLoader

    class Cubes:
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

    n, bs = 203, 8                                         # per-rank batch 8: global batches of 32 / 64 + a ragged tail of 11

    def run(rank, w, b):
        torch.manual_seed(5); random.seed(5)
        return [bt[0] for bt in DeviceLoader(Cubes(n), batch_size=b, shuffle=True, rank=rank, world=w)]

    single = run(0, 1, bs * world)                         # the reference's single process at the global batch size
    shards = [run(r, world, bs) for r in range(world)]
    assert all(len(s) == len(single) for s in shards) and len(single) == n // (bs * world) + 1
    for bi, glob in enumerate(single):
        per = len(glob) // world
        got = [i for r in range(world) for i in shards[r][bi]]
        assert got == glob[:per * world], (bi, len(glob))
        assert all(len(shards[r][bi]) == per for r in range(world))


@pytest.mark.parametrize("world,fault,field", [(2, "", None), (3, "", None), (8, "", None), (2, "order", "same_collective_order"),
                                               (8, "grad", "same_reduced_gradients"), (2, "grid", "same_grid_sequence")])
def test_bench_verify_detects_divergent_ranks(world, fault, field):
    """`bench.py --verify`: the self-check a multi-rank run carries in its JSON line.  On gloo ranks with
    synthetic gradients: consistent ranks answer dp_consistent = true with the bucket list; a rank that reports another
    collective order, holds a different reduced gradient, or drew another grid turns it false and names what differed."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env["OMP_NUM_THREADS"] = "1"
    env["HSIMAE_DRYRUN_FAULT"] = fault
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-run", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    v = out["dp_verify"]
    assert v["ranks"] == world and len(v["buckets"]) >= 2
    assert v["matches_single_collective"] is (fault != "grad")      # (the perturbed rank also disagrees with the plain collective)
    if not fault:
        assert out["dp_consistent"] is True and all(v[k] for k in ("same_reduced_gradients", "same_grid_sequence", "same_collective_order"))
    else:
        assert out["dp_consistent"] is False and v[field] is False
        assert all(v[k] for k in ("same_reduced_gradients", "same_grid_sequence", "same_collective_order") if k != field)


def test_verify_step_flags_a_bucketed_result_that_differs_from_the_plain_collective():
    """The other leg of the check, single process: a bucket that was reduced before its kernels had written it shows up as a
    mismatch against the one-collective reference even when every rank agrees with every other."""
    from hsimae_amd.parallel import flat_hash, verify_step
    g = torch.Generator().manual_seed(0)
    ref = torch.randn(50_000, generator=g)
    good = ref + 1e-8 * ref.abs().max() * torch.randn(50_000, generator=g)        # summation-order noise
    assert verify_step(good, ref, [(3, 9)], [(0, 50_000)], 1.0)["dp_consistent"] is True
    bad = good.clone(); bad[1234] = 0.0
    v = verify_step(bad, ref, [(3, 9)], [(0, 50_000)], 1.0)
    assert v["dp_consistent"] is False and v["matches_single_collective"] is False
    a = torch.arange(1000, dtype=torch.float32); b = a.clone(); b[[3, 7]] = b[[7, 3]]
    assert flat_hash(a) == flat_hash(a.clone()) and flat_hash(a) != flat_hash(b)


def test_bench_watchdog_prints_the_line_when_the_legs_after_the_timed_region_hang():
    """`bench.py`'s rank 0 owes the driver ONE JSON line.  The comm / self-check / replay legs run after the timed region; if one
    hangs (a collective another rank never entered), the watchdog prints the line with what it holds and ends the process
    with a NON-ZERO status (the line is incomplete and the process has used the GPU: rc 0 would tell the driver the run was whole)."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "line = {'metric': 'm', 'value': 1.0, 'comm': {'buckets': 5}}\n"
            "w = bench._Watchdog(0.3, line)\n"
            "time.sleep(30)\n"                      # the hung leg
            "print('never')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1 and "never" not in r.stdout
    d = json.loads(out[0])
    assert d["value"] == 1.0 and "did not finish" in d["watchdog"] and "comm" in d["watchdog"]
    # a SLOW leg (not a hung one) that keeps adding keys while the timer fires: a line still comes out
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "line = {'metric': 'm', 'value': 2.0}\n"
            "w = bench._Watchdog(0.3, line)\n"
            "t0 = time.time(); i = 0\n"
            "while time.time() - t0 < 20:\n"
            "    line['k%%d' %% (i %% 5000)] = i; i += 1\n"
            "    if i %% 5000 == 0: [line.pop('k%%d' %% j) for j in range(5000)]\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 3 and len(out) == 1 and json.loads(out[0])["value"] == 2.0 and "watchdog" in json.loads(out[0])
    # the normal end: finish() prints once, the timer is cancelled, nothing follows
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "w = bench._Watchdog(0.3, {'metric': 'm'}); w.line['extra'] = 1; w.finish(); w.finish(); time.sleep(0.8)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0 and len(out) == 1 and json.loads(out[0]) == {"metric": "m", "extra": 1}
