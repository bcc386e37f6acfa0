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
    def __init__(self, process_group=None, bucket_bytes=4 << 20):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._works = []
        self._pending = None
        self._cb = None
        self._flat = None

    def broadcast(self, tensor, src=0):
        if self.world_size > 1:
            dist.broadcast(tensor, src=src, group=self.group)

    # -- called from inside hsimae_backward (host thread, after the kernels of the range were enqueued)
    def _on_range(self, stage, off, ln, user):
        lo, hi = self._pending if self._pending is not None else (off + ln, off + ln)
        if off + ln != lo:                      # not contiguous with what is pending: flush first
            self._launch(lo, hi)
            lo, hi = off + ln, off + ln
        lo = off
        if hi - lo >= self.bucket_elems:
            self._launch(lo, hi)
            self._pending = None
        else:
            self._pending = (lo, hi)

    def _launch(self, lo, hi):
        if hi > lo and self.world_size > 1:
            self._works.append(dist.all_reduce(self._flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def make_callback(self, flat_grad):
        self._flat, self._works, self._pending = flat_grad, [], None
        self._cb = _lib.BUCKET_CB(self._on_range)       # keep a reference for the duration of the call
        return self._cb

    def finish(self):
        if self._pending is not None:
            self._launch(*self._pending)
            self._pending = None
        for w in self._works:
            w.wait()                                    # compute stream waits for the communication stream
        self._works = []

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
