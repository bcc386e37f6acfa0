"""Host-side pieces of the training loop mirror (rows N1 schedule, N4 checkpoint) that need no GPU."""
import math
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hsimae_amd.checkpoint import rng_state, save_final, set_rng_state  # noqa: E402
from hsimae_amd.sched import CosineLRScheduler  # noqa: E402


class Opt:
    def __init__(self):
        self.param_groups = [{"lr": 5e-3, "weight_decay": 5e-2}, {"lr": 5e-3, "weight_decay": 0.0}]


def test_cosine_schedule_semantics_as_used_by_the_reference_loop():
    """timm-0.9 single-cycle cosine + linear warm-up (parity unpinned: timm is not installable here)."""
    iters = 200
    o = Opt()
    s = CosineLRScheduler(o, t_initial=iters, lr_min=1e-6, warmup_t=int(np.ceil(iters * 0.05)))
    assert [g["lr"] for g in o.param_groups] == [0.0, 0.0]           # lr is warmup_lr_init until the first step(t)
    seen = []
    for t in range(iters + 3):                                         # loop order: optimizer.step(); scheduler.step(t)
        seen.append(o.param_groups[0]["lr"])
        s.step(t)
    assert seen[0] == 0.0 and seen[1] == 0.0                           # the first two optimizer steps run at lr 0
    assert abs(seen[6] - 5e-3 * 5 / 10) < 1e-12                        # linear warm-up over 10 iterations
    assert abs(seen[101] - (1e-6 + 0.5 * (5e-3 - 1e-6) * (1 + math.cos(math.pi * 100 / 200)))) < 1e-12
    assert seen[-1] == 1e-6 and max(seen) <= 5e-3
    assert all(a >= b for a, b in zip(seen[12:], seen[13:]))           # monotone decay after warm-up
    o2 = Opt()
    s2 = CosineLRScheduler(o2, t_initial=iters, lr_min=1e-6, warmup_t=10)
    s.step(57)
    s2.load_state_dict(s.state_dict())
    assert o2.param_groups[1]["lr"] == o.param_groups[1]["lr"]


def test_rng_state_roundtrip_and_final_files(tmp_path):
    random.seed(3); np.random.seed(3); torch.manual_seed(3)
    st = rng_state()
    a = (random.random(), float(np.random.rand()), torch.rand(1).item())
    set_rng_state(st)
    assert a == (random.random(), float(np.random.rand()), torch.rand(1).item())
    m = torch.nn.Linear(3, 2)
    save_final(m, str(tmp_path), "m.pkl", [1.5, 1.25], [])
    sd = torch.load(os.path.join(tmp_path, "m.pkl"))
    assert list(sd) == ["weight", "bias"] and sd["weight"].dtype == torch.float32
    log = np.load(os.path.join(tmp_path, "train_log.npy"), allow_pickle=True)
    assert list(log[0]) == [1.5, 1.25] and len(log[1]) == 0           # [epoch_loss_list, val_loss_list] as the reference saves
