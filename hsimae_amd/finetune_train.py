"""The reference's fine-tuning entry point (Model_Finetuning.py:66-240 `dual_branch_finetuning`) on the MI355X-native
parts: same signature, same order of operations, hyper-parameters and RNG consumption per step.

  model      : hsimae_amd.DualViT (row N3), optionally initialised from a pretraining checkpoint by key (:84-96)
  data       : labeled / unlabeled / validation cubes resident in HBM (`HSIdataset` below, the reference's :26-63 with the
               same two python-`random` flip draws per training sample), batched by hsimae_amd.data.DeviceLoader
  step       : loss = lamda * loss_rec + CrossEntropy(ignore_index=0)(class_pred, y)             (:150-160)
  optimizer  : hsimae_amd.FusedAdamW (default betas), CosineLRScheduler stepped per EPOCH with
               t_initial=epochs, lr_min=lr/100, warmup_t=ceil(0.1 epochs), warmup_lr_init=lr/100  (:103-106, 236)
  metrics    : OA / AA / kappa on the labeled pixels (gt != 0, classes shifted by one)             (:171-178, 206-215)
Not carried over: the matplotlib figure (:131-137, 222-233, 240-241).  The reference's defaults dim=144 / dec_dim=72 (widths that
are not multiples of the kernels' 32-deep k-step) run zero-padded to 160 / 96 inside the library.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch

from .data import DeviceLoader
from .finetune import DualViT, HSIViT
from .optim import FusedAdamW
from .pretrain import seed_everything
from .sched import CosineLRScheduler


class HSIdataset:
    """Model_Finetuning.py:26-63: a list of [h, w, Bands] cubes (+ labels); training samples are flipped along w then h
    with probability 0.5 each (python `random`, horizontal draw first); items come out as [1, Bands, h, w] fp32."""

    def __init__(self, data_list, gt=None, train=False, device="cuda:0"):
        self.device = torch.device(device)
        arr = np.ascontiguousarray(np.stack([np.asarray(d, dtype=np.float32) for d in data_list]))
        self._x = torch.from_numpy(arr).to(self.device)                         # [n, h, w, Bands]
        self.gt = None if gt is None else np.asarray(gt)
        self._y = None if gt is None else torch.from_numpy(self.gt.astype(np.int64)).to(self.device)
        self.train = train

    def __len__(self):
        return self._x.shape[0]

    def batch(self, indices):
        idx = torch.as_tensor(np.asarray(indices, dtype=np.int64)).to(self.device)
        x = self._x[idx]
        if self.train:
            fh = np.zeros(len(indices), dtype=bool); fv = np.zeros(len(indices), dtype=bool)
            for i in range(len(indices)):
                fh[i] = random.random() < 0.5                                   # np.flip(data, 1): along w
                fv[i] = random.random() < 0.5                                   # np.flip(data, 0): along h
            fh_t, fv_t = torch.from_numpy(fh).to(self.device), torch.from_numpy(fv).to(self.device)
            x = torch.where(fh_t.view(-1, 1, 1, 1), x.flip(2), x)
            x = torch.where(fv_t.view(-1, 1, 1, 1), x.flip(1), x)
        x = x.permute(0, 3, 1, 2).unsqueeze(1)                                  # [n, 1, Bands, h, w] (band-fastest view)
        return x if self._y is None else (x, self._y[idx])


def spilt_dataset(data, label, training_ratio=0.8):
    """Utils/Preprocessing.py:276-300: per-class split after one np.random.permutation; the first
    (1 - ratio) * count samples of each class (in shuffled order) go to validation."""
    label = np.asarray(label)
    shuffled = np.random.permutation(np.arange(label.shape[0]))
    n_classes = len(np.unique(label))
    assert n_classes == label.max()
    val_quota = np.array([np.sum(label == c + 1) for c in range(n_classes)]) * (1 - training_ratio)
    seen = np.zeros(n_classes)
    tr, va = [], []
    for i in shuffled:
        c = label[i] - 1
        seen[c] += 1
        (va if seen[c] <= val_quota[c] else tr).append(i)
    if training_ratio == 1:
        va = tr[:int(len(tr) * 0.2)]
    return [data[i] for i in tr], label[tr], [data[i] for i in va], label[va]


def scores(gt, pred):
    """(OA, AA, kappa, per-class recall) over gt != 0 with classes shifted by one (Model_Finetuning.py:171-178):
    sklearn's accuracy_score / recall_score(average=None) / cohen_kappa_score restated on the confusion matrix."""
    gt, pred = np.asarray(gt).astype(np.int64), np.asarray(pred).astype(np.int64)
    keep = gt != 0
    g, p = gt[keep] - 1, pred[keep] - 1
    labels = np.unique(np.concatenate([g, p]))
    lut = {v: i for i, v in enumerate(labels)}
    cm = np.zeros((len(labels), len(labels)), dtype=np.float64)
    for a, b in zip(g, p):
        cm[lut[a], lut[b]] += 1
    n = cm.sum()
    oa = np.trace(cm) / n
    present = cm.sum(1) > 0
    ca = np.diag(cm)[present] / cm.sum(1)[present]                             # recall of the classes present in gt
    pe = float((cm.sum(0) * cm.sum(1)).sum()) / (n * n)
    kappa = (oa - pe) / (1 - pe) if pe < 1 else 0.0
    return float(oa), float(ca.mean()), float(kappa), ca


def dual_branch_finetuning(data_list, labeled_index, unlabeled_data, gt, save_dir, model_name, pretrained=None,
                           lr=1e-3, wd=5e-3, depth=12, dim=144, dec_depth=2, dec_dim=72, s_depth=6,
                           epochs=100, mask_ratio=0.5, lamda=5, batch_size=32, device="cuda:0", log=print):
    device = torch.device(device)
    h, w, c = data_list[0].shape
    n_class = int(np.max(gt) + 1)
    model = DualViT(img_size=h, patch_size=3, in_chans=1, bands=c, b_patch_size=8, num_class=n_class,
                    embed_dim=dim, depth=depth, num_heads=dim // 16, s_depth=s_depth,
                    decoder_embed_dim=dec_dim, decoder_depth=dec_depth, decoder_num_heads=dec_dim // 8,
                    norm_pix_loss=True, trunc_init=True, drop_path=0.2).to(device)
    save_path = os.path.join(save_dir, model_name.replace(".pkl", ""))
    os.makedirs(save_path, exist_ok=True)
    if pretrained:
        model_dict = model.state_dict()
        loaded = torch.load(pretrained, map_location=device)
        model_dict.update({k: v for k, v in loaded.items() if k in model_dict})
        model.load_state_dict(model_dict)

    optimizer = FusedAdamW(model, lr=lr, weight_decay=wd)
    scheduler = CosineLRScheduler(optimizer, t_initial=epochs, lr_min=lr * 0.01, warmup_t=int(np.ceil(0.1 * epochs)),
                                  warmup_lr_init=lr * 0.01)
    criterion = torch.nn.CrossEntropyLoss(reduction="mean", ignore_index=0)

    data_arr = [data_list[i] for i in labeled_index]
    tr_x, tr_y, va_x, va_y = spilt_dataset(data_arr, gt, training_ratio=0.5)
    train_ds = HSIdataset(tr_x, tr_y, train=True, device=device)
    unl_ds = HSIdataset(unlabeled_data, train=True, device=device)
    val_ds = HSIdataset(va_x, va_y, device=device)
    train_dl = DeviceLoader(train_ds, batch_size=batch_size, shuffle=True)
    unl_bs = int(np.ceil(len(unl_ds) / len(train_dl)) / 2)
    unl_dl = DeviceLoader(unl_ds, batch_size=unl_bs, shuffle=True)
    val_dl = DeviceLoader(val_ds, batch_size=512, shuffle=False)
    log(f"train {len(train_ds)} labeled / {len(unl_ds)} unlabeled cubes, {len(train_dl)} iterations per epoch")

    epoch_loss_list, val_loss_list, val_value = [], [], None
    for epoch in range(epochs):
        model.train()
        seed_everything(42 + epoch); labeled_iter = iter(train_dl)              # `stable(loader, 42 + epoch)` twice
        seed_everything(42 + epoch); unlabeled_iter = iter(unl_dl)
        train_loss, preds, gts = 0.0, [], []
        for _ in range(len(train_dl)):
            x, y = next(labeled_iter)
            x_u = next(unlabeled_iter)
            loss_rec, _, _, outputs = model(x, x_u, mask_ratio=mask_ratio)
            loss = lamda * loss_rec + criterion(outputs, y)
            preds.append(outputs.detach().argmax(1)); gts.append(y)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            train_loss += loss.item()
        epoch_loss_list.append(train_loss / len(train_dl))

        model.eval()
        with torch.no_grad():
            seed_everything(42 + epoch)
            val_loss, preds, gts = 0.0, [], []
            for x, y in val_dl:
                outputs = model(x, mask_ratio=mask_ratio)
                val_loss += criterion(outputs, y).item()
                preds.append(outputs.argmax(1)); gts.append(y)
        val_value = list(scores(torch.cat(gts).cpu().numpy(), torch.cat(preds).cpu().numpy()))
        val_loss_list.append(val_loss / len(val_dl))
        scheduler.step(epoch)

    torch.save(model.state_dict(), os.path.join(save_dir, model_name))
    return val_value, epoch_loss_list, val_loss_list


def test_model(data_cubes, test_gt, gt, save_dir, model_name, depth=12, dim=96, s_depth=6, device="cuda:0"):
    """Model_Finetuning.test_model (:243-300): every cube of the scene through HSIViT loaded key-filtered from the
    fine-tuned checkpoint, class = 1 + argmax over logits[:, 1:], scores on the labeled test pixels.
    -> (oa, aa, kappa, per-class recall, prediction map shaped like `gt`).  The colour-map PNGs (:296-298) are not written."""
    device = torch.device(device)
    h, w, c = data_cubes[0].shape
    n_class = int(np.max(gt) + 1)
    model = HSIViT(img_size=h, patch_size=3, in_chans=1, bands=c, b_patch_size=8, num_class=n_class, embed_dim=dim, depth=depth,
                   num_heads=dim // 16, s_depth=s_depth, sep_pos_embed=True, use_learnable_pos_emb=False).to(device)
    model_dict = model.state_dict()
    loaded = torch.load(os.path.join(save_dir, model_name), map_location=device)
    model_dict.update({k: v for k, v in loaded.items() if k in model_dict})
    model.load_state_dict(model_dict)
    model.eval()
    dataset = HSIdataset(data_cubes, device=device)
    preds = []
    with torch.no_grad():
        for x in DeviceLoader(dataset, batch_size=256, shuffle=False):
            preds.append(model(x)[:, 1:].argmax(1))
    pred = (torch.cat(preds).cpu().numpy() + 1).reshape(np.asarray(gt).shape)
    pred_all = pred.copy()
    pred[np.asarray(gt) == 0] = 0
    oa, aa, kappa, ca = scores(np.asarray(test_gt).reshape(-1), pred.reshape(-1))
    return oa, aa, kappa, ca, pred_all
