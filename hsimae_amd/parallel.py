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
