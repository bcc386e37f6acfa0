#!/usr/bin/env python3
"""Golden fixture for the input pipeline from the REFERENCE implementation.

Runs only in the build container (needs /root/reference).  It imports the reference's `Model_Pretraining`
(for `HSIdataset4PT`), `Utils.Preprocessing.get_split_info` and `Utils.Seed_Everything.stable`, and iterates a real
`torch.utils.data.DataLoader` exactly as the training loop does (Model_Pretraining.py:75, 92-95).  Only inputs and
outputs are recorded.

`Model_Pretraining` imports `timm.scheduler.CosineLRScheduler` at module scope for the LR schedule; timm is not in
this image.  An empty placeholder module is registered so that the import statement succeeds — nothing on the
dataset path touches it.

    python tests/golden/make_golden_loader.py        ->  tests/golden/loader.npz
"""
import contextlib
import io
import os
import random
import sys
import types

import numpy as np
import torch
from torch.utils.data import DataLoader

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
if "timm" not in sys.modules:
    timm = types.ModuleType("timm")
    sched = types.ModuleType("timm.scheduler")
    sched.CosineLRScheduler = None
    timm.scheduler = sched
    sys.modules["timm"], sys.modules["timm.scheduler"] = timm, sched
with contextlib.redirect_stdout(io.StringIO()):
    import Model_Pretraining as MP  # noqa: E402
    from Utils.Preprocessing import get_split_info  # noqa: E402
    from Utils.Seed_Everything import stable  # noqa: E402


def main():
    rng = np.random.default_rng(7)
    out = {}
    for tag, dtype, (mx, mn) in (("f32", np.float32, (1, 0)), ("f64", np.float64, (7, -3))):
        scenes = [rng.standard_normal((14, 16, 16)).astype(dtype), rng.standard_normal((12, 13, 16)).astype(dtype) * 3]
        cut = []
        for num, sc in enumerate(scenes):
            cut += get_split_info(sc, (9, 9, sc.shape[2]), (3, 3, 1), num, mx, mn)
        cut = np.array(cut, dtype=np.int16)
        out[f"{tag}_scene0"], out[f"{tag}_scene1"], out[f"{tag}_cut"] = scenes[0], scenes[1], cut
        ds = MP.HSIdataset4PT([scenes, cut], train=True)
        dl = DataLoader(ds, batch_size=5, shuffle=True, num_workers=0, pin_memory=False)
        for epoch in range(2):
            batches = [x.numpy() for x in stable(dl, 42 + epoch)]
            out[f"{tag}_epoch{epoch}"] = np.concatenate(batches, 0)
            # where the two RNG streams stand after the epoch (pins how much of each the loader consumed)
            out[f"{tag}_epoch{epoch}_next_random"] = np.array([random.random()])
            out[f"{tag}_epoch{epoch}_next_torch"] = torch.rand(1).numpy()
        ev = MP.HSIdataset4PT([scenes, cut], train=False)
        out[f"{tag}_eval3"] = ev[3].numpy()
    np.savez_compressed(os.path.join(HERE, "loader.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
