"""hsimae_amd — MI355X-native (gfx950) HSIMAE masked-autoencoder pretraining path.

`HSIMAE` mirrors the reference `Models.HSIMAE` module surface; the arithmetic lives in
`libhsimae_hip.so` (hand-written HIP kernels, C ABI in include/hsimae_hip.h).
"""
from .model import HSIMAE, swiglu_hidden, sincos_table  # noqa: F401
from .optim import FusedAdamW  # noqa: F401
from .data import HSIdataset4PT, DeviceLoader  # noqa: F401
from .sched import CosineLRScheduler  # noqa: F401
from .pretrain import mask_pretraining  # noqa: F401
from .finetune import DualViT, HSIViT  # noqa: F401
from .finetune_train import dual_branch_finetuning, test_model  # noqa: F401

__version__ = "0.1.0"
