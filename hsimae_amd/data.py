"""Device-resident input pipeline: the reference's `HSIdataset4PT` + `DataLoader` pair
(Model_Pretraining.py:21-54, 75, 92-95) with the scenes kept in HBM and one HIP launch per batch.

Same constructor, same sample semantics, same RNG consumption:
  * `HSIdataset4PT(data_cubes, train, device)`: `data_cubes = [scenes, cut_info]` as returned by the reference's
    `get_data_cut_file` (Utils/Preprocessing.py:82-117); `ds[i]` is `[1, Bands, 9, 9]` fp32.
  * flips draw python `random.random()` twice per sample, horizontal first (Model_Pretraining.py:28-38, 47-48);
  * `DeviceLoader(ds, batch_size, shuffle=True)` iterates like `DataLoader(..., shuffle=True, num_workers=0)`: one
    int64 seed from torch's default generator per epoch, `randperm` with a private generator, last batch kept.
The arithmetic runs in `hsimae_cube_gather` (csrc/loader.hip); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import random

import numpy as np
import torch

from . import _lib


def draw_base_seed() -> None:
    """`DataLoader.__iter__` draws the iterator's `_base_seed` (one int64 from the default generator, used only by worker
    processes) when the iterator is CREATED — before any batch is requested."""
    torch.empty((), dtype=torch.int64).random_()


def sampler_order(n: int) -> list[int]:
    """torch.utils.data.RandomSampler (replacement=False, generator=None): its seed is drawn from the default generator
    when the FIRST batch is requested, then `randperm` runs on a private generator."""
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).tolist()


def random_sampler_order(n: int) -> list[int]:
    """Index order of one epoch when nothing else touches the default generator between `iter()` and the first
    `next()`: base seed, then the sampler's seed and permutation."""
    draw_base_seed()
    return sampler_order(n)


def draw_flips(n: int, train: bool) -> np.ndarray:
    """uint8 per sample, bit 0 = flip along w (np.flip(data, 1)), bit 1 = flip along h (np.flip(data, 0))."""
    out = np.zeros(n, dtype=np.uint8)
    if train:
        for i in range(n):
            h = random.random() < 0.5
            v = random.random() < 0.5
            out[i] = (1 if h else 0) | (2 if v else 0)
    return out


def build_tables(scenes, cut_info):
    """Concatenated scene buffer + per-scene element offsets / widths + validated int16 cut table (host side)."""
    if len(scenes) == 0:
        raise ValueError("no scenes")
    dt = scenes[0].dtype
    if dt not in (np.float32, np.float64):
        raise TypeError(f"scenes must be float32 or float64 arrays (got {dt}): the reference's numpy arithmetic "
                        "is reproduced bit-exactly only for those")
    bands = scenes[0].shape[2]
    off, widths, flat, cur = [], [], [], 0
    for s in scenes:
        if s.ndim != 3 or s.dtype != dt or s.shape[2] != bands:
            raise ValueError("all scenes must be [h, w, bands] arrays of one dtype and band count")
        off.append(cur)
        widths.append(s.shape[1])
        flat.append(np.ascontiguousarray(s).reshape(-1))
        cur += s.size
    cut = np.asarray(cut_info)
    if cut.ndim != 2 or cut.shape[1] != 6:
        raise ValueError("cut_info must be [n, 6] rows (c, h, w, scene, max, min)")
    cut = cut.astype(np.int16, copy=False)
    for c, h, w, num, mx, mn in cut:
        if not (0 <= num < len(scenes)) or h < 0 or w < 0 or h + 9 > scenes[num].shape[0] or w + 9 > scenes[num].shape[1]:
            raise ValueError(f"cut row {(c, h, w, num)} does not fit its scene")
        if mx == mn:
            raise ValueError("cut row with max == min")
    return np.concatenate(flat), np.asarray(off, dtype=np.int64), np.asarray(widths, dtype=np.int32), cut, bands


class HSIdataset4PT:
    def __init__(self, data_cubes, train=False, device="cuda:0"):
        self.train = train
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("hsimae_amd.data.HSIdataset4PT is the device-resident loader: it needs a GPU")
        flat, off, widths, cut, bands = build_tables(data_cubes[0], data_cubes[1])
        self.bands = bands
        self.f64 = flat.dtype == np.float64
        self.cut_info = cut
        self._scenes = torch.from_numpy(flat).to(self.device)
        self._off = torch.from_numpy(off).to(self.device)
        self._w = torch.from_numpy(widths).to(self.device)
        self._cut = torch.from_numpy(cut.copy()).to(self.device)

    def __len__(self):
        return len(self.cut_info)

    def gather(self, indices, flips=None, band_fastest=True) -> torch.Tensor:
        """[N, 1, Bands, 9, 9] fp32 for the given cut rows.  `flips` (uint8 per sample) defaults to none.
        band_fastest=True returns the reference's own memory order (a permuted view of [N, 9, 9, Bands])."""
        n = len(indices)
        idx = torch.as_tensor(np.asarray(indices, dtype=np.int64)).to(self.device)
        fl = None if flips is None else torch.as_tensor(np.asarray(flips, dtype=np.uint8)).to(self.device)
        B = self.bands
        if band_fastest:
            buf = torch.empty(n, 9, 9, B, dtype=torch.float32, device=self.device)
            out = buf.permute(0, 3, 1, 2).unsqueeze(1)
        else:
            out = torch.empty(n, 1, B, 9, 9, dtype=torch.float32, device=self.device)
        p = _lib.CubeParams(scenes=self._scenes.data_ptr(), scene_f64=int(self.f64), scene_off=self._off.data_ptr(),
                            scene_w=self._w.data_ptr(), bands=B, cut=self._cut.data_ptr(), index=idx.data_ptr(),
                            flips=_lib.ptr(fl), N=n, out=out.data_ptr(), sn=out.stride(0), sb=out.stride(2),
                            sh=out.stride(3), sw=out.stride(4))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.load().hsimae_cube_gather(C.byref(p), stream), "hsimae_cube_gather")
        return out

    def batch(self, indices) -> torch.Tensor:
        """The batch a DataLoader(num_workers=0) would collate for these indices (draws the flips now)."""
        return self.gather(indices, draw_flips(len(indices), self.train))

    def __getitem__(self, index):
        return self.batch([index])[0]


class DeviceLoader:
    """`DataLoader(dataset, batch_size=bs, shuffle=True, num_workers=0)` (Model_Pretraining.py:75) on the device."""

    def __init__(self, dataset: HSIdataset4PT, batch_size=1, shuffle=False, drop_last=False, rank=0, world=1):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        if not 0 <= rank < world:
            raise ValueError("rank must be in [0, world)")
        self.rank, self.world = rank, world               # data parallel: per-rank batch_size, see _LoaderIter

    def __len__(self):
        n, gb = len(self.dataset), self.batch_size * self.world
        if self.drop_last:
            return n // gb
        # the ragged last global batch is kept only if every rank gets at least one cube (see _LoaderIter)
        return n // gb + (1 if n % gb >= self.world else 0)

    def __iter__(self):
        return _LoaderIter(self)


class _LoaderIter:
    """One epoch.  RNG consumption follows `_SingleProcessDataLoaderIter`: the base seed is drawn here, at `iter()`; the
    sampler's seed and permutation at the first `next()` — so two loaders created back to back and then advanced in turn
    (Model_Finetuning.py:142-149) see the default generator in the reference's order.

    `rank` / `world` (data parallel, not in the reference): every rank draws the SAME permutation (same seeds) and takes
    the rank-th contiguous slice of each global batch of `batch_size * world` indices, so the union over ranks is the
    reference's single-process batch sequence at a global batch of `batch_size * world`."""

    def __init__(self, loader):
        self.loader = loader
        self.order = None
        self.pos = 0
        draw_base_seed()

    def __iter__(self):
        return self

    def __next__(self):
        ld = self.loader
        n = len(ld.dataset)
        if self.order is None:
            self.order = sampler_order(n) if ld.shuffle else list(range(n))
        gb = ld.batch_size * ld.world
        if self.pos >= n:
            raise StopIteration
        idx = self.order[self.pos:self.pos + gb]
        if ld.drop_last and len(idx) < gb:
            raise StopIteration
        self.pos += gb
        if ld.world > 1:
            # every rank consumes the flips' python-random draws of the WHOLE global batch, in the reference's order, and
            # keeps its own slice: the python-random stream (which also picks the masking grid) stays identical on all ranks.
            # Equal shares only (equal sum(mask) per rank makes the mean of rank means the global mean): the last, ragged
            # global batch drops its < world leftover cubes.
            flips = draw_flips(len(idx), ld.dataset.train)
            per = len(idx) // ld.world
            if per == 0:
                raise StopIteration
            sl = slice(ld.rank * per, (ld.rank + 1) * per)
            return ld.dataset.gather(idx[sl], flips[sl])
        return ld.dataset.batch(idx)
