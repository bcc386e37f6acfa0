"""Data-parallel gradient averaging for HSIMAE pretraining: one process per GPU, RCCL over xGMI.

The reference is single-process; pretraining shards naturally by sample, so the only
exchange step is the gradient all-reduce.  The backward schedule in libhsimae_hip.so reports each group of
parameters whose gradients are complete (reverse registration order => contiguous suffixes of the flat
gradient buffer); this reducer merges them into ~bucket_bytes buckets and launches an asynchronous
all-reduce per bucket from the compute stream, so communication overlaps the remaining backward kernels.
Gradients arrive pre-scaled by 1/world (folded into dLoss/dpred), so SUM == mean.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib


def plan_buckets(ranges, bucket_elems):
    """Merge (off, len) ranges arriving back-to-front into contiguous buckets of >= bucket_elems elements.
    Returns a list of (off, len, trigger_index): launch the bucket after range `trigger_index` arrives."""
    out, lo, hi = [], None, None
    for i, (off, ln) in enumerate(ranges):
        if ln <= 0:
            continue
        if lo is None:
            lo, hi = off, off + ln
        elif off + ln == lo:
            lo = off
        elif off == hi:
            hi = off + ln
        else:                                   # non-adjacent: flush what we have
            out.append((lo, hi - lo, i - 1))
            lo, hi = off, off + ln
        if hi - lo >= bucket_elems:
            out.append((lo, hi - lo, i))
            lo = hi = None
    if lo is not None:
        out.append((lo, hi - lo, len(ranges) - 1))
    return out


class GradReducer:
    def __init__(self, process_group=None, bucket_bytes=4 << 20, force_collectives=False):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.force_collectives = force_collectives       # issue the all-reduces with a single rank too (path test)
        self._works = []
        self._pending = []                                # disjoint [lo, hi) intervals waiting to reach bucket size
        self._cb = None
        self._flat = None
        self._error = None
        self._launch_stream = None
        self.launched = []                                # (lo, hi) of every collective of the last backward (tests)

    def broadcast(self, tensor, src=0):
        if self.world_size > 1:
            dist.broadcast(tensor, src=src, group=self.group)

    def launch_stream_handle(self, device):
        """Raw handle of the stream the collectives are launched from (hsimae_io.bucket_stream): the library makes it wait,
        through events, for the kernels of each range it reports — including those on its side stream — so a range can be
        reduced while the compute stream goes on with the next block."""
        if self._launch_stream is None or self._launch_stream.device != device:
            self._launch_stream = torch.cuda.Stream(device=device)
        return self._launch_stream.cuda_stream

    # -- called from inside hsimae_backward (host thread, after the kernels of the range were enqueued).
    # ctypes swallows exceptions raised in a callback: they are stashed and re-raised from finish().
    def _on_range(self, stage, off, ln, user):
        if self._error is not None or ln <= 0:
            return
        try:
            self._add(off, off + ln)
        except BaseException as e:                        # noqa: BLE001 - must not propagate into the C caller
            self._error = e

    def _add(self, lo, hi):
        merged = True
        while merged:                                     # merge with neighbours (ranges arrive back to front, two stacks interleaved)
            merged = False
            for iv in self._pending:
                if iv[1] == lo or iv[0] == hi:
                    lo, hi = min(lo, iv[0]), max(hi, iv[1])
                    self._pending.remove(iv)
                    merged = True
                    break
        if hi - lo >= self.bucket_elems:
            self._launch(lo, hi)
        else:
            self._pending.append((lo, hi))

    def _launch(self, lo, hi):
        if hi <= lo:
            return
        self.launched.append((lo, hi))
        if self.world_size > 1 or self.force_collectives:
            t = self._flat[lo:hi]
            if self._launch_stream is not None and t.is_cuda:
                with torch.cuda.stream(self._launch_stream):
                    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append(w)

    def make_callback(self, flat_grad):
        self._flat, self._works, self._pending, self._error, self.launched = flat_grad, [], [], None, []
        if self._launch_stream is not None and flat_grad.is_cuda:
            # the zero-fill of the gradient buffer (and anything else queued so far) precedes every collective
            self._launch_stream.wait_stream(torch.cuda.current_stream(flat_grad.device))
        self._cb = _lib.BUCKET_CB(self._on_range)       # keep a reference for the duration of the call
        return self._cb

    def finish(self):
        err = self._error
        if err is None:
            try:
                for lo, hi in sorted(self._pending, reverse=True):
                    self._launch(lo, hi)
            except BaseException as e:                    # noqa: BLE001
                err = e
        self._pending = []
        for w in self._works:
            w.wait()                                    # compute stream waits for the communication stream
        self._works = []
        self._error = None
        if err is not None:
            raise RuntimeError("gradient all-reduce failed inside the backward pass; ranks may have diverged") from err

    # -- device-agnostic helper used by the CPU (gloo) tests and by callers that own a plain flat buffer
    def reduce_ranges(self, flat, ranges):
        """All-reduce(sum) `flat` bucket by bucket following `plan_buckets`; every element exactly once."""
        works = []
        for lo, ln, _ in plan_buckets(ranges, self.bucket_elems):
            if self.world_size > 1:
                works.append(dist.all_reduce(flat[lo:lo + ln], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        return flat


# ---------------------------------------------------------------------------------------------------------------------------
# Self-check of a data-parallel step (bench.py --verify; VERDICT r04 "Next round" 5).  No 8-GPU node has run this path yet:
# the first multi-rank run must DETECT a wrong bucket order or a collective that ran ahead of the kernels feeding it, not just
# time the step.  Device-agnostic (the CPU tests drive it over gloo).
def flat_hash(t: torch.Tensor) -> int:
    """64-bit position-weighted hash of a tensor's BITS (fp32 / int32 / int64), computed where the tensor lives: equal for
    bit-identical buffers, different (up to 2^-64 collisions) when any element or any two elements' positions differ."""
    v = t.detach().reshape(-1)
    if v.dtype == torch.float32:
        v = v.view(torch.int32)
    v = v.to(torch.int64)
    n = v.numel()
    if n == 0:
        return 0
    idx = torch.arange(n, device=v.device, dtype=torch.int64)
    w = idx * 2 + 1                                          # odd weights: multiplication is a bijection mod 2^64
    mixed = (v + 0x1F3D5B79) * w
    mixed = mixed ^ (mixed >> 29)
    return int(((mixed * 0x2545F4914F6CDD1D) + idx).sum().item())       # int64 arithmetic wraps mod 2^64


def _hash_ints(values) -> int:
    h = 1469598103934665603
    for x in values:
        h = ((h ^ (int(x) & 0xFFFFFFFFFFFFFFFF)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h - (1 << 64) if h >= (1 << 63) else h


def verify_step(flat_bucketed: torch.Tensor, flat_reference: torch.Tensor | None, grids, launched, loss: float,
                group=None, rtol: float = 2e-6) -> dict:
    """Cross-rank consistency of ONE data-parallel step (every rank calls it with its own view):

      * `flat_bucketed`  — this rank's flat gradient buffer after the bucketed, backward-overlapped all-reduce;
      * `flat_reference` — the same step's LOCAL gradients (reducer detached) summed over ranks by ONE plain all-reduce after a
        full device synchronise and divided by world: what the bucketed path must equal up to the summation order inside the
        collective (`rtol` of the largest magnitude).  None skips this leg;
      * `grids`          — the (len_t, len_l) sequence this rank drew; `launched` — the (lo, hi) of the collectives it issued, in
        issue order.

    All-gathers [hash(flat_bucketed), hash(grids), hash(launched), #launched] over the group.  `dp_consistent` is True iff every
    rank reduced to bit-identical gradients, drew the same grids, issued the same collectives in the same order, and (when
    given) the bucketed result matches the single-collective reference on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    dev = flat_bucketed.device
    mine = torch.tensor([flat_hash(flat_bucketed), _hash_ints([v for g in grids for v in g]),
                         _hash_ints([v for r in launched for v in r]), len(launched)], dtype=torch.int64, device=dev)
    ref_err = None
    if flat_reference is not None:
        scale = float(flat_reference.abs().max().item()) or 1.0
        ref_err = float((flat_bucketed - flat_reference).abs().max().item()) / scale
    ok_local = torch.tensor([1 if (ref_err is None or ref_err <= rtol) else 0], dtype=torch.int64, device=dev)
    if world > 1:
        rows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine, group=group)
        dist.all_reduce(ok_local, op=dist.ReduceOp.MIN, group=group)
        table = torch.stack(rows).cpu()
    else:
        table = mine.cpu()[None]
    same = [bool((table[:, c] == table[0, c]).all()) for c in range(4)]
    return {"dp_consistent": bool(all(same) and int(ok_local.item()) == 1),
            "same_reduced_gradients": same[0], "same_grid_sequence": same[1], "same_collective_order": same[2] and same[3],
            "matches_single_collective": None if ref_err is None else bool(int(ok_local.item()) == 1),
            "single_collective_max_rel_err_rank0": ref_err, "ranks": world,
            "gradient_hash": f"{int(table[0, 0]) & 0xFFFFFFFFFFFFFFFF:016x}",
            "buckets": [[int(lo), int(hi)] for lo, hi in launched], "loss_rank0": loss}
