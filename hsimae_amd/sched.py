"""LR schedule of the reference training loop (second half): `timm.scheduler.CosineLRScheduler`
as `Model_Pretraining.py:88,104` uses it — `CosineLRScheduler(optimizer, t_initial=iters, lr_min=1e-6,
warmup_t=ceil(0.05 iters))`, stepped once per iteration with `scheduler.step(iter_num)` AFTER `optimizer.step()`.

PARITY UNPINNED: timm (0.9.12 in the reference's requirements) is not installable in the build image, so there is no
fixture; the semantics below are timm 0.9's single-cycle cosine with linear warm-up as recorded in SURVEY.md 8c:
  * construction sets every group's lr to `warmup_lr_init` (0) when `warmup_t > 0`;
  * `step(t)`: `t < warmup_t` -> `warmup_lr_init + t (base - warmup_lr_init) / warmup_t`;
               otherwise    -> `lr_min + (base - lr_min) (1 + cos(pi t / t_initial)) / 2` for `t < t_initial`
               (no warm-up prefix: the cosine is evaluated at t, not t - warmup_t), `lr_min` afterwards.
Because the loop steps the schedule after the optimizer with t = iter_num starting at 0, the first two optimizer
steps run at lr 0.  Works with `torch.optim` optimizers and `FusedAdamW` (anything with `param_groups`)."""
from __future__ import annotations

import math


class CosineLRScheduler:
    def __init__(self, optimizer, t_initial, lr_min=0.0, warmup_t=0, warmup_lr_init=0.0):
        self.optimizer = optimizer
        self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init = int(t_initial), lr_min, int(warmup_t), warmup_lr_init
        self.base_values = [g["lr"] for g in optimizer.param_groups]
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        if self.warmup_t:
            self._set([self.warmup_lr_init] * len(self.base_values))
        self.last_t = -1

    def _set(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v

    def get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * (b - self.warmup_lr_init) / self.warmup_t for b in self.base_values]
        if t < self.t_initial:
            return [self.lr_min + 0.5 * (b - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial)) for b in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def step(self, t):
        self.last_t = int(t)
        self._set(self.get_lr(t))

    def state_dict(self):
        return {"last_t": self.last_t, "base_values": list(self.base_values)}

    def load_state_dict(self, sd):
        self.last_t, self.base_values = int(sd["last_t"]), list(sd["base_values"])
        if self.last_t >= 0:
            self._set(self.get_lr(self.last_t))
