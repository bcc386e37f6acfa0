"""The reference's entry point `mask_pretraining` (Model_Pretraining.py:57-113) on the native parts: loader (N2) ->
model -> FusedAdamW + cosine schedule (N1) -> files / resume (N4)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loader_oracle as LO  # noqa: E402

pytestmark = pytest.mark.gpu


def cubes():
    rng = np.random.default_rng(5)
    scenes = [rng.random((16, 17, 32)).astype(np.float32), rng.random((13, 15, 32)).astype(np.float32)]
    cut = []
    for num, sc in enumerate(scenes):
        cut += LO.split_info(sc.shape, (9, 9, 32), (3, 3, 1), num, 1, 0)
    return [scenes, np.array(cut, dtype=np.int16)]


KW = dict(img_size=9, bands=32, mask_ratio=0.5, lr=5e-3, wd=5e-2, bs=8, depth=3, dim=32, s_depth=2, dec_dim=32, dec_depth=2,
          log=lambda *_: None)


def test_mask_pretraining_files_and_resume(tmp_path):
    import hsimae_amd
    from hsimae_amd.pretrain import seed_everything

    def mask_pretraining(*a, **k):
        seed_everything(0)               # weight init draws from the global streams (the reference script seeds once, too)
        return hsimae_amd.mask_pretraining(*a, **k)
    d1, d2 = str(tmp_path / "a"), str(tmp_path / "b")
    model, losses = mask_pretraining(cubes(), d1, "m.pkl", epochs=3, **KW)
    assert len(losses) == 3 and all(np.isfinite(losses)) and losses[-1] < losses[0]
    sd = torch.load(os.path.join(d1, "m.pkl"))
    ref = model.state_dict()
    assert list(sd) == list(ref) and all(v.dtype == torch.float32 and v.shape == ref[k].shape for k, v in sd.items())
    log = np.load(os.path.join(d1, "train_log.npy"), allow_pickle=True)
    assert np.allclose(np.array(list(log[0]), dtype=np.float64), losses)
    # interrupted after 2 of 3 epochs, then resumed: same trajectory (atomics reorder the last bits of the gradients)
    ck = os.path.join(d2, "resume.pt")
    os.makedirs(d2)
    _, l2 = mask_pretraining(cubes(), d2, "m.pkl", epochs=2, resume_path=ck, **KW)
    assert os.path.exists(ck) and len(l2) == 2
    # the 2-epoch run has a different schedule length (t_initial): resume it as a 2-epoch run and check nothing is redone,
    _, l2b = mask_pretraining(cubes(), d2, "m.pkl", epochs=2, resume_path=ck, **KW)
    assert l2b == l2
    # and a 3-epoch run interrupted after its 2nd epoch continues onto the uninterrupted trajectory
    ck3 = os.path.join(d2, "resume3.pt")
    import hsimae_amd.pretrain as P
    orig = P.save_final
    try:
        P.save_final = lambda *a, **k: None
        calls = {"n": 0}
        real_save = P.save_resume

        def stop_after_two(*a, **k):
            real_save(*a, **k)
            calls["n"] += 1
            if calls["n"] == 2:
                raise KeyboardInterrupt
        P.save_resume = stop_after_two
        with pytest.raises(KeyboardInterrupt):
            mask_pretraining(cubes(), d2, "m3.pkl", epochs=3, resume_path=ck3, **KW)
        P.save_resume = real_save
    finally:
        P.save_final = orig
        P.save_resume = real_save
    _, l3 = mask_pretraining(cubes(), d2, "m3.pkl", epochs=3, resume_path=ck3, **KW)
    assert len(l3) == 3
    assert np.allclose(l3, losses, rtol=2e-3), (l3, losses)
