"""Row N2 on the GPU: hsimae_cube_gather (csrc/loader.hip) through hsimae_amd.data against the fixture recorded from
the reference's HSIdataset4PT + DataLoader and against the oracle on larger synthetic scenes.  Bit-exact."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loader_oracle as LO  # noqa: E402

pytestmark = pytest.mark.gpu
FX = np.load(os.path.join(ROOT, "tests", "golden", "loader.npz"))


def seed_all(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_device_loader_reproduces_reference_epochs(tag):
    from hsimae_amd.data import DeviceLoader, HSIdataset4PT
    scenes, cut = [FX[f"{tag}_scene0"], FX[f"{tag}_scene1"]], FX[f"{tag}_cut"]
    ds = HSIdataset4PT([scenes, cut], train=True)
    dl = DeviceLoader(ds, batch_size=5, shuffle=True)
    assert len(ds) == 18 and len(dl) == 4
    for ep in range(2):
        seed_all(42 + ep)
        got = torch.cat([b.cpu() for b in dl], 0)
        assert got.shape == (18, 1, 16, 9, 9) and got.dtype == torch.float32
        assert np.array_equal(got.numpy(), FX[f"{tag}_epoch{ep}"])
        assert random.random() == float(FX[f"{tag}_epoch{ep}_next_random"][0])
        assert torch.rand(1).item() == float(FX[f"{tag}_epoch{ep}_next_torch"][0])
    ev = HSIdataset4PT([scenes, cut], train=False)
    assert np.array_equal(ev[3].cpu().numpy(), FX[f"{tag}_eval3"])
    # the two output layouts hold the same values; the default is the reference's band-fastest memory order
    a = ev.gather([0, 5, 17], flips=[3, 1, 2])
    b = ev.gather([0, 5, 17], flips=[3, 1, 2], band_fastest=False)
    assert a.stride()[2] == 1 and b.is_contiguous() and torch.equal(a, b)


@pytest.mark.parametrize("dtype,mxmn,bands", [(np.float32, (1, 0), 96), (np.float32, (4000, -12), 96), (np.float64, (9, 2), 48),
                                              (np.float32, (3, 1), 20)])
def test_gather_against_oracle_large(dtype, mxmn, bands):
    from hsimae_amd.data import HSIdataset4PT
    rng = np.random.default_rng(11)
    scenes = [(rng.standard_normal((40, 37, bands)) * 50).astype(dtype), (rng.standard_normal((23, 61, bands)) * 900).astype(dtype),
              rng.random((9, 9, bands)).astype(dtype)]
    cut = []
    for num, sc in enumerate(scenes):
        cut += LO.split_info(sc.shape, (9, 9, bands), (3, 3, 1), num, *mxmn)
    cut = np.array(cut, dtype=np.int16)
    ds = HSIdataset4PT([scenes, cut], train=True)
    idx = rng.integers(0, len(cut), 300).tolist() + [len(cut) - 1, 0]
    flips = rng.integers(0, 4, len(idx)).astype(np.uint8)
    got = ds.gather(idx, flips).cpu()
    want = torch.stack([LO.getitem(scenes, cut, j, bool(f & 1), bool(f & 2)) for j, f in zip(idx, flips)], 0)
    assert torch.equal(got, want)
    assert ds.gather([], None).shape == (0, 1, bands, 9, 9)


def test_loader_output_feeds_the_model_like_a_contiguous_batch():
    """The band-fastest view goes straight into HSIMAE.forward (patch gather takes strides): same loss as a copy."""
    from hsimae_amd import HSIMAE
    from hsimae_amd.data import HSIdataset4PT
    rng = np.random.default_rng(3)
    scenes = [rng.random((30, 30, 48)).astype(np.float32)]
    cut = np.array(LO.split_info(scenes[0].shape, (9, 9, 48), (3, 3, 1), 0, 1, 0), dtype=np.int16)
    ds = HSIdataset4PT([scenes, cut], train=True)
    x = ds.gather(list(range(32)), np.arange(32, dtype=np.uint8) % 4)
    torch.manual_seed(0)
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=48, b_patch_size=8, embed_dim=128, depth=12, num_heads=8, s_depth=9,
               decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True).cuda()
    g = torch.Generator().manual_seed(5)
    noise = (torch.rand(32, 6, generator=g), torch.rand(32, 9, generator=g))
    la = m(x, 0.75, noise=noise, grid=(2, 7))[0].item()
    lb = m(x.contiguous(), 0.75, noise=noise, grid=(2, 7))[0].item()
    assert la == lb
