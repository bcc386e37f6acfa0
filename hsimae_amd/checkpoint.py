"""Checkpoint I/O and resume.  The reference saves only the final `state_dict` and
`train_log.npy` (Model_Pretraining.py:111-113) and cannot resume; `save_final` writes exactly those two files (same
keys / shapes / dtype, so `Model_Finetuning.py:85-96` loads them), and `save_resume` / `load_resume` add what a
multi-hour run needs to continue bit-for-bit on the host side: optimizer moments, schedule position, epoch /
iteration counters, the loss history and the four RNG streams the loop consumes (python `random`, numpy, torch CPU,
torch device)."""
from __future__ import annotations

import os
import random

import numpy as np
import torch


def save_final(model, save_path, model_name, epoch_loss_list, val_loss_list=()):
    os.makedirs(save_path, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(save_path, model_name))
    np.save(os.path.join(save_path, "train_log.npy"), np.array([list(epoch_loss_list), list(val_loss_list)], dtype=object)
            if len(epoch_loss_list) != len(val_loss_list) else np.array([epoch_loss_list, val_loss_list]))


def rng_state(device=None):
    st = {"python": random.getstate(), "numpy": np.random.get_state(), "torch": torch.get_rng_state()}
    if device is not None and torch.device(device).type == "cuda":
        st["cuda"] = torch.cuda.get_rng_state(device)
    return st


def set_rng_state(st, device=None):
    random.setstate(st["python"])
    np.random.set_state(st["numpy"])
    torch.set_rng_state(st["torch"])
    if "cuda" in st and device is not None:
        torch.cuda.set_rng_state(st["cuda"], device)


def save_resume(path, model, optimizer, scheduler, epoch, iter_num, epoch_loss_list, device=None):
    """Atomic (write + rename) so an interrupted save never leaves a half-written file behind."""
    blob = {"model": model.state_dict(), "optimizer": optimizer.state_dict(),
            "scheduler": scheduler.state_dict() if scheduler is not None else None,
            "epoch": int(epoch), "iter_num": int(iter_num), "epoch_loss_list": list(epoch_loss_list),
            "rng": rng_state(device)}
    tmp = path + ".tmp"
    torch.save(blob, tmp)
    os.replace(tmp, path)


def load_resume(path, model, optimizer, scheduler, device=None):
    blob = torch.load(path, map_location="cpu", weights_only=False)      # RNG states must stay CPU byte tensors
    model.load_state_dict(blob["model"])
    optimizer.load_state_dict(blob["optimizer"])
    if scheduler is not None and blob["scheduler"] is not None:
        scheduler.load_state_dict(blob["scheduler"])
    set_rng_state(blob["rng"], device)
    return blob["epoch"], blob["iter_num"], list(blob["epoch_loss_list"])
