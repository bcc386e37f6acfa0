"""The reference's training entry point (Model_Pretraining.py:57-113 `mask_pretraining`) on the MI355X-native parts:
same signature, same order of operations and RNG consumption, same output files — plus `resume_path` (row N4).

  dataset / loader : hsimae_amd.data.HSIdataset4PT + DeviceLoader   (row N2, scenes resident in HBM)
  model            : hsimae_amd.HSIMAE                               (rows a-1 … a-14)
  optimizer        : hsimae_amd.FusedAdamW, CosineLRScheduler        (row N1; schedule parity unpinned, see sched.py)
  files            : <save_path>/<model_name> (state_dict), <save_path>/train_log.npy   (row N4)
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch

from .checkpoint import load_resume, save_final, save_resume
from .data import DeviceLoader, HSIdataset4PT
from .model import HSIMAE
from .optim import FusedAdamW
from .sched import CosineLRScheduler


def seed_everything(seed):
    """Utils/Seed_Everything.py:7-16 (the cudnn switches have no counterpart here)."""
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def mask_pretraining(data_cubes, save_path, model_name, img_size=9, bands=32, mask_ratio=0.50, lr=5e-3, wd=5e-2, bs=512,
                     epochs=100, depth=12, dim=64, s_depth=6, dec_dim=48, dec_depth=2, resume_path=None, device="cuda:0",
                     log=print):
    device = torch.device(device)
    train_dataset = HSIdataset4PT(data_cubes, train=True, device=device)
    del data_cubes
    log(f"dataset load finished: {len(train_dataset)} cubes; net {[dim, depth, dec_dim, dec_depth]}")

    model = HSIMAE(img_size=img_size, patch_size=3, in_chans=1, bands=bands, b_patch_size=8,
                   embed_dim=dim, depth=depth, num_heads=dim // 16, s_depth=s_depth,
                   decoder_embed_dim=dec_dim, decoder_depth=dec_depth, decoder_num_heads=dec_dim // 8,
                   norm_pix_loss=True, trunc_init=True).to(device)
    os.makedirs(save_path, exist_ok=True)

    train_dataload = DeviceLoader(train_dataset, batch_size=bs, shuffle=True)
    optimizer = FusedAdamW(model, lr=lr, weight_decay=wd, betas=(0.9, 0.95))
    iters = epochs * len(train_dataload)
    scheduler = CosineLRScheduler(optimizer, t_initial=iters, lr_min=1e-6, warmup_t=int(np.ceil(iters * 0.05)))

    epoch_loss_list, val_loss_list = [], []
    iter_num, first_epoch = 0, 0
    if resume_path is not None and os.path.exists(resume_path):
        model._ensure_flat(device)                      # the optimizer state is laid out like the flat buffer
        first_epoch, iter_num, epoch_loss_list = load_resume(resume_path, model, optimizer, scheduler, device)
        log(f"resumed at epoch {first_epoch}, iteration {iter_num}")
    for epoch in range(first_epoch, epochs):
        train_loss = 0.0
        model.train()
        seed_everything(42 + epoch)                     # `for x in stable(train_dataload, 42 + epoch)`
        for x in train_dataload:
            loss, _, _ = model(x, mask_ratio=mask_ratio)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            scheduler.step(iter_num)
            iter_num += 1
            train_loss += loss.item()
        epoch_loss_list.append(train_loss / len(train_dataload))
        if resume_path is not None:
            save_resume(resume_path, model, optimizer, scheduler, epoch + 1, iter_num, epoch_loss_list, device)
    save_final(model, save_path, model_name, epoch_loss_list, val_loss_list)
    return model, epoch_loss_list
