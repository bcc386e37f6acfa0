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


def dp_context(device):
    """(rank, world, device) of this process.  Under `torch.distributed.run` (RANK / WORLD_SIZE / LOCAL_RANK in the
    environment, or an initialised process group) the job is data parallel, one process per GPU, RCCL: the process group
    is created here if the launcher has not done it, and the device becomes cuda:LOCAL_RANK."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if dist.is_available() and dist.is_initialized():
        # the launcher made the group: still one GPU per rank (RCCL refuses ranks that share a device)
        if torch.device(device).type == "cuda" and dist.get_world_size() > 1 and dist.get_backend() == "nccl":
            local = int(os.environ.get("LOCAL_RANK", str(dist.get_rank() % max(1, torch.cuda.device_count()))))
            device = torch.device("cuda", local if torch.cuda.device_count() > local else 0)
            torch.cuda.set_device(device)
        return dist.get_rank(), dist.get_world_size(), device
    if world <= 1:
        return 0, 1, device
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    backend = os.environ.get("HSIMAE_DP_BACKEND")       # tests: "gloo" lets two ranks share one GPU (RCCL refuses that)
    if torch.device(device).type == "cuda":
        device = torch.device("cuda", local if torch.cuda.device_count() > local else 0)
        torch.cuda.set_device(device)
        if (backend or "nccl") == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        dist.init_process_group(backend or "gloo", rank=rank, world_size=world)
    return rank, world, device


def mask_pretraining(data_cubes, save_path, model_name, img_size=9, bands=32, mask_ratio=0.50, lr=5e-3, wd=5e-2, bs=512,
                     epochs=100, depth=12, dim=64, s_depth=6, dec_dim=48, dec_depth=2, resume_path=None, device="cuda:0",
                     log=print):
    """`bs` is the GLOBAL batch, as in the reference; under a data-parallel launch each rank takes bs / world cubes of every
    batch (same permutation and python-random stream on every rank), gradients are averaged with the bucketed RCCL
    all-reduce overlapped with the backward (hsimae_amd/parallel.py), and rank 0 writes the files."""
    device = torch.device(device)
    rank, world, device = dp_context(device)
    if bs % world:
        raise ValueError(f"global batch {bs} is not divisible by the {world} ranks")
    train_dataset = HSIdataset4PT(data_cubes, train=True, device=device)
    del data_cubes
    log(f"dataset load finished: {len(train_dataset)} cubes; net {[dim, depth, dec_dim, dec_depth]}")

    model = HSIMAE(img_size=img_size, patch_size=3, in_chans=1, bands=bands, b_patch_size=8,
                   embed_dim=dim, depth=depth, num_heads=dim // 16, s_depth=s_depth,
                   decoder_embed_dim=dec_dim, decoder_depth=dec_depth, decoder_num_heads=dec_dim // 8,
                   norm_pix_loss=True, trunc_init=True).to(device)
    if world > 1:
        model.enable_data_parallel()                    # broadcasts rank 0's initial weights
    if rank == 0:
        os.makedirs(save_path, exist_ok=True)

    train_dataload = DeviceLoader(train_dataset, batch_size=bs // world, shuffle=True, rank=rank, world=world)
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
        T = bands // 8
        for x in train_dataload:
            noise = None
            if world > 1:
                # the masking noise of the GLOBAL batch, drawn identically on every rank (same device-generator seed), each
                # rank keeping its rows: the ranks' samples get different noise, as the samples of one big batch would
                per = x.shape[0]
                n1 = torch.rand(per * world, T, device=device)
                n2 = torch.rand(per * world, 9, device=device)
                noise = (n1[rank * per:(rank + 1) * per], n2[rank * per:(rank + 1) * per])
            loss, _, _ = model(x, mask_ratio=mask_ratio, noise=noise)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            scheduler.step(iter_num)
            iter_num += 1
            train_loss += loss.item()
        epoch_loss = train_loss / len(train_dataload)
        if world > 1:                                   # logging only: mean of the ranks' epoch means
            import torch.distributed as dist
            t = torch.tensor([epoch_loss], dtype=torch.float64, device=device)
            dist.all_reduce(t)
            epoch_loss = float(t.item()) / world
        epoch_loss_list.append(epoch_loss)
        if resume_path is not None and rank == 0:
            save_resume(resume_path, model, optimizer, scheduler, epoch + 1, iter_num, epoch_loss_list, device)
    if rank == 0:
        save_final(model, save_path, model_name, epoch_loss_list, val_loss_list)
    return model, epoch_loss_list
