"""CPU restatement of the reference's pretraining input pipeline.  TEST INFRASTRUCTURE ONLY:
imported by tests/ (and nothing in the product path); parity PINNED against tests/golden/loader.npz, which was
produced by running the reference's own `HSIdataset4PT` + `DataLoader` (tests/golden/make_golden_loader.py).

Follows
  * `HSIdataset4PT.__getitem__`      /root/reference/Model_Pretraining.py:40-51
  * `random_horizontal_filp` / `random_vertical_filp`   Model_Pretraining.py:28-38 (python `random`, H first, then V)
  * `get_split_info`                 /root/reference/Utils/Preprocessing.py:69-79 (cut table rows (c, h, w, scene, max, min))
  * the loop `for x in stable(train_dataload, 42 + epoch)`   Model_Pretraining.py:92-95 with
    `DataLoader(batch_size=bs, shuffle=True, num_workers=0)` (:75): torch's RandomSampler draws one int64 seed from the
    default generator when the iterator is created and permutes with a private generator; batches are fetched
    lazily, sample by sample, so the python-`random` stream is consumed 2 draws per sample in batch order.
"""
from __future__ import annotations

import random
from itertools import product

import numpy as np
import torch


def initial_seq(length: int, size: int, stride: int) -> np.ndarray:
    """Window starts along one axis (Utils/Preprocessing.py:8-20).  `stride` is the number of steps per window
    length (step = size // stride); the count formula is the reference's, and the last start is forced flush
    with the end of the axis."""
    whole = length // size
    rest = length - whole * size
    step = int(size // stride)
    extra = rest // step
    left = rest - extra * step
    count = int((whole - 1) * stride + extra + (1 if left == 0 else 2))
    seq = np.arange(0, count * step, step)
    seq[-1] = length - size
    return seq


def split_info(shape, target_size, stride, num, mx, mn):
    """Utils/Preprocessing.py:69-79: product(ch_seq, row_seq, col_seq, [num], [max], [min])."""
    w, h, c = shape
    ws, hs, cs = stride
    rowsize, colsize, chsize = target_size
    ch_seq = initial_seq(c, chsize, cs)
    row_seq = initial_seq(w, rowsize, ws)
    col_seq = initial_seq(h, colsize, hs)
    return list(product(ch_seq, row_seq, col_seq, [num], [mx], [mn]))


def getitem(scenes, cut, index: int, flip_h: bool, flip_v: bool) -> torch.Tensor:
    """One sample [1, B, 9, 9] fp32 (Model_Pretraining.py:40-51); the flips are given, not drawn."""
    c, h, w, num, mx, mn = cut[index]
    data = scenes[num][h:h + 9, w:w + 9, :]
    data = (data - mn) / (mx - mn)
    if flip_h:
        data = np.flip(data, 1)
    if flip_v:
        data = np.flip(data, 0)
    t = torch.tensor(data.copy(), dtype=torch.float32)
    return t.unsqueeze(0).permute(0, 3, 1, 2)


def draw_flips(n: int, train: bool = True):
    """The 2n python-random draws a batch of n samples consumes: per sample H (`< 0.5`) then V."""
    out = []
    for _ in range(n):
        if train:
            fh = random.random() < 0.5
            fv = random.random() < 0.5
        else:
            fh = fv = False
        out.append((fh, fv))
    return out


def sampler_order(n: int) -> list[int]:
    """torch.utils.data.RandomSampler.__iter__ (replacement=False, generator=None), n <= 2**31."""
    # DataLoader.__iter__ first draws the iterator's `_base_seed` (one int64 from the default generator, used only
    # by worker processes), then the sampler draws its own seed when the first batch is requested
    torch.empty((), dtype=torch.int64).random_()
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).tolist()


def epoch(scenes, cut, batch_size: int, train: bool = True):
    """Yields (indices, flips, batch[N,1,B,9,9]) exactly as iterating the reference DataLoader would."""
    order = sampler_order(len(cut))
    for i in range(0, len(order), batch_size):
        idx = order[i:i + batch_size]
        fl = draw_flips(len(idx), train)
        yield idx, fl, torch.stack([getitem(scenes, cut, j, a, b) for j, (a, b) in zip(idx, fl)], 0)
