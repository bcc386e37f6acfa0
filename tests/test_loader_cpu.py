"""Row N2 (input pipeline) on CPU: the oracle restatement and the host-side planning logic of hsimae_amd.data against
the fixture recorded from the reference's own HSIdataset4PT + DataLoader (tests/golden/make_golden_loader.py)."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loader_oracle as LO  # noqa: E402

FX = np.load(os.path.join(ROOT, "tests", "golden", "loader.npz"))


def seed_all(seed):          # what the reference's `stable(loader, seed)` does to the two streams the loader uses
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def scenes_cut(tag):
    return [FX[f"{tag}_scene0"], FX[f"{tag}_scene1"]], FX[f"{tag}_cut"]


@pytest.mark.parametrize("tag,mxmn", [("f32", (1, 0)), ("f64", (7, -3))])
def test_cut_table_restatement(tag, mxmn):
    scenes, cut = scenes_cut(tag)
    rows = []
    for num, sc in enumerate(scenes):
        rows += LO.split_info(sc.shape, (9, 9, sc.shape[2]), (3, 3, 1), num, *mxmn)
    assert np.array_equal(np.array(rows, dtype=np.int16), cut)
    # window starts: flush with the end of the axis, step 3
    assert list(LO.initial_seq(14, 9, 3)) == [0, 3, 5] and list(LO.initial_seq(16, 9, 3)) == [0, 3, 6, 7]


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_oracle_epochs_bit_exact_and_rng_positions(tag):
    scenes, cut = scenes_cut(tag)
    for ep in range(2):
        seed_all(42 + ep)
        got = torch.cat([b for _, _, b in LO.epoch(scenes, cut, 5, train=True)], 0).numpy()
        want = FX[f"{tag}_epoch{ep}"]
        assert got.dtype == np.float32 and got.shape == want.shape
        assert np.array_equal(got, want)                                  # bit-exact, flips and order included
        assert random.random() == float(FX[f"{tag}_epoch{ep}_next_random"][0])     # 2 draws per sample, no more
        assert torch.rand(1).item() == float(FX[f"{tag}_epoch{ep}_next_torch"][0])
    ev = LO.getitem(scenes, cut, 3, False, False).numpy()
    assert np.array_equal(ev, FX[f"{tag}_eval3"])


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_host_planning_of_device_loader_matches_reference(tag):
    """random_sampler_order / draw_flips / build_tables are what DeviceLoader feeds the HIP kernel with; checked here
    with the oracle doing the arithmetic the kernel does on the GPU (tests/test_gpu_loader.py)."""
    from hsimae_amd import data as D
    scenes, cut = scenes_cut(tag)
    flat, off, widths, cut16, bands = D.build_tables(scenes, cut)
    assert bands == 16 and flat.dtype == scenes[0].dtype and list(off) == [0, scenes[0].size] and list(widths) == [16, 13]
    assert np.array_equal(flat[off[1]:].reshape(scenes[1].shape), scenes[1])
    seed_all(42)
    order = D.random_sampler_order(len(cut16))
    rows = []
    for i in range(0, len(order), 5):
        idx = order[i:i + 5]
        fl = D.draw_flips(len(idx), True)
        rows += [LO.getitem(scenes, cut16, j, bool(f & 1), bool(f & 2)) for j, f in zip(idx, fl)]
    assert np.array_equal(torch.stack(rows, 0).numpy(), FX[f"{tag}_epoch0"])
    assert not D.draw_flips(4, False).any()
    with pytest.raises(TypeError):
        D.build_tables([scenes[0].astype(np.int16)], cut)
    bad = cut.copy(); bad[0, 1] = 30
    with pytest.raises(ValueError):
        D.build_tables(scenes, bad)


def test_device_loader_refuses_cpu():
    from hsimae_amd import data as D
    scenes, cut = scenes_cut("f32")
    with pytest.raises(RuntimeError):
        D.HSIdataset4PT([scenes, cut], train=True, device="cpu")
