"""Fused AdamW for `hsimae_amd.HSIMAE`.

The reference builds `torch.optim.AdamW` over two name-filtered parameter groups (Model_Pretraining.py:80-86) and
steps it once per iteration (:102).  With fwd+bwd at ~30 ms, 535 per-tensor updates are pure launch overhead; here
the whole flat parameter buffer is updated by ONE kernel (`hsimae_adamw_step`) with torch's AdamW arithmetic, and
the model is told to refresh its packed bf16 weight images.  `param_groups` is kept (one dict per reference group,
sharing `lr`) so LR schedulers that write `group['lr']` keep working."""
from __future__ import annotations

import torch

from . import _lib


class FusedAdamW:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, no_decay=("bias", "norm"), strict=True):
        self.model = model
        self.strict = bool(strict)                   # see _sync_grads
        self._mask_cache = {}
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        decay, nodecay = [], []
        self._groups_of = []
        # Parameters that do not live in the model's flat buffer (DualViT's cls_head) are stepped by a stock
        # torch.optim.AdamW with the same hyper-parameters; their lr follows param_groups[0] / [1].
        in_flat = {id(p) for p in model._plist()} if hasattr(model, "_plist") else None
        extra_decay, extra_nodecay = [], []
        for n, p in model.named_parameters():
            is_nd = any(k in n for k in no_decay)
            if in_flat is not None and id(p) not in in_flat:
                if p.requires_grad:
                    (extra_nodecay if is_nd else extra_decay).append(p)
                    (nodecay if is_nd else decay).append(p)
                continue
            if not p.requires_grad or n == "mask_token":        # mask_token never receives a gradient
                self._groups_of.append(2)
            elif is_nd:
                self._groups_of.append(1); nodecay.append(p)
            else:
                self._groups_of.append(0); decay.append(p)
        self.param_groups = [dict(params=decay, lr=lr, weight_decay=weight_decay, betas=tuple(betas), eps=eps),
                             dict(params=nodecay, lr=lr, weight_decay=0.0, betas=tuple(betas), eps=eps)]
        self._extra = None
        if extra_decay or extra_nodecay:
            groups = [(0, dict(params=extra_decay, weight_decay=weight_decay)), (1, dict(params=extra_nodecay, weight_decay=0.0))]
            groups = [(i, g) for i, g in groups if g["params"]]
            self._extra_of = [i for i, _ in groups]          # which reference group each extra group follows
            self._extra = torch.optim.AdamW([g for _, g in groups], lr=lr, betas=tuple(betas), eps=eps)
        self.step_count = 0
        self._flat_id = None
        self.exp_avg = self.exp_avg_sq = self._group = None

    def _bind(self):
        m = self.model
        if m._flat is None:
            raise RuntimeError("FusedAdamW: run a forward pass on the GPU first (the flat parameter buffer does not exist yet)")
        if self._flat_id != m._flat.data_ptr():
            flat = m._flat
            if self.exp_avg is None or self.exp_avg.numel() != flat.numel():
                self.exp_avg = torch.zeros_like(flat)
                self.exp_avg_sq = torch.zeros_like(flat)
            else:
                self.exp_avg, self.exp_avg_sq = self.exp_avg.to(flat.device), self.exp_avg_sq.to(flat.device)
            grp = torch.empty(flat.numel(), dtype=torch.uint8)
            for off, size, gid in zip(m._offs, m._sizes, self._groups_of):
                grp[off:off + size] = gid
            self._group_host = grp
            self._group = grp.to(flat.device)
            self._mask_cache = {}
            self._flat_id = flat.data_ptr()

    def zero_grad(self, set_to_none: bool = True):
        zg = getattr(self.model, "zero_grad", None)
        if zg is not None:
            zg(set_to_none=set_to_none)              # HSIMAE.zero_grad also records that no gradient is at home any more
            return
        for p in self.model.parameters():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _sync_grads(self):
        """`.grad` is the contract, the flat gradient buffer only its usual home.  A parameter whose `.grad` is None (cleared
        by zero_grad and not touched by this step's backward — e.g. the encoder after a decoder-only backward) is SKIPPED like
        torch.optim.AdamW skips it: it is masked out of this step (group id 2: no update, no weight decay, moments kept), so a
        stale range of the flat buffer is never applied.  A `.grad` that is some other tensor (assigned or accumulated by the
        caller) is copied into its flat view first.  Returns the group-id tensor to use for this step.

        Cost: the model keeps track of which of its two parameter ranges (encoder, decoder) had their `.grad` attached by a
        backward since the last `zero_grad` (`HSIMAE._grads_home`).  When both were — every step of the reference's loop —
        the default (`strict=True`) compares every `.grad` with its flat view (532 attribute reads, ~50 us of host time that
        overlaps the device's backward) and the step is ONE launch: a `.grad` set to None or replaced by another tensor after
        the backward (clipping by assignment, masking, a hook) is seen, exactly as `torch.optim.AdamW` would see it.
        `strict=False` is the opt-in fast path for loops that never touch `.grad` between backward and step: four identity
        comparisons (first / last parameter of each range); a middle parameter's replaced `.grad` is then NOT noticed.  Only when
        something is missing or foreign is the per-parameter walk done, and its mask is built on the host and uploaded once
        (cached per pattern) — no per-parameter device writes."""
        m = self.model
        params, views = m._params_cache, m._grad_views
        home = getattr(m, "_grads_home", None)
        if home is not None and home[0] and home[1]:
            if self.strict:
                if all(params[i].grad is views[i] for i in m._trainable):
                    return self._group
            else:
                probe = (m._enc_idx[0], m._enc_idx[-1], m._dec_idx[0], m._dec_idx[-1])
                if all(params[i].grad is views[i] for i in probe):
                    return self._group
        missing, foreign = [], []
        for i in m._trainable:
            g = params[i].grad
            if g is None:
                missing.append(i)
            elif g is not views[i] and (g.data_ptr() != views[i].data_ptr() or g.shape != views[i].shape):
                foreign.append(i)
        for i in foreign:
            views[i].copy_(params[i].grad.to(views[i].dtype).reshape(views[i].shape))
        if not missing:
            return self._group
        key = tuple(missing)
        grp = self._mask_cache.get(key)
        if grp is None:
            host = self._group_host.clone()
            hv = host.numpy()
            for i in missing:
                hv[m._offs[i]: m._offs[i] + m._sizes[i]] = 2
            grp = host.to(self._group.device)             # one upload per pattern
            if len(self._mask_cache) >= 4:
                self._mask_cache.pop(next(iter(self._mask_cache)))
            self._mask_cache[key] = grp
        return grp

    @torch.no_grad()
    def step(self):
        self._bind()
        m = self.model
        g0 = self.param_groups[0]
        self.step_count += 1
        b1, b2 = g0["betas"]
        stream = torch.cuda.current_stream(m._flat.device).cuda_stream
        group = self._sync_grads()
        _lib.check(_lib.load().hsimae_adamw_step(
            m._flat.data_ptr(), m._flat_grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            group.data_ptr(), m._flat.numel(), float(g0["lr"]), float(b1), float(b2), float(g0["eps"]),
            float(g0["weight_decay"]), self.step_count, stream), "hsimae_adamw_step")
        m._packed_version = -1                    # packed bf16 images are stale now
        if self._extra is not None:
            for ge, gi in zip(self._extra.param_groups, self._extra_of):
                ge["lr"] = self.param_groups[gi]["lr"]
            self._extra.step()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "extra": self._extra.state_dict() if self._extra is not None else None}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg, self.exp_avg_sq = sd["exp_avg"], sd["exp_avg_sq"]
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
        if self._extra is not None and sd.get("extra") is not None:
            self._extra.load_state_dict(sd["extra"])
        self._flat_id = None
