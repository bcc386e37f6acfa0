"""hsimae_amd — MI355X-native (gfx950) HSIMAE masked-autoencoder pretraining path.

`HSIMAE` mirrors the reference `Models.HSIMAE` module surface; the arithmetic lives in
`libhsimae_hip.so` (hand-written HIP kernels, C ABI in include/hsimae_hip.h).
"""
import os as _os

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With the default, a process that has
# created an RCCL communicator — even one it never uses — runs this library's step 0.7 ms (4 %) slower on MI355X: the
# communicator's idle streams share hardware queues with the compute streams (scripts/exp_ddp_slow.sh, profiles/r04_ddp_queues.txt:
# 17.07 vs 16.31 ms; with 8 — or 2 — queues 16.30).  The variable is read when HIP initialises, so it is set at import, before
# any GPU call of a normal program; an explicit setting of the user's wins, HSIMAE_KEEP_HW_QUEUES=1 leaves the default alone.
if not _os.environ.get("HSIMAE_KEEP_HW_QUEUES"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .model import HSIMAE, swiglu_hidden, sincos_table  # noqa: F401
from .optim import FusedAdamW  # noqa: F401
from .data import HSIdataset4PT, DeviceLoader  # noqa: F401
from .sched import CosineLRScheduler  # noqa: F401
from .pretrain import mask_pretraining  # noqa: F401
from .finetune import DualViT, HSIViT  # noqa: F401
from .finetune_train import dual_branch_finetuning, test_model  # noqa: F401

__version__ = "0.1.0"
