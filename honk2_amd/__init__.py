"""honk2_amd: MI355X-native implementation of honk2's batched keyword-spotting inference path.

wav -> MFCC front end -> res8/res15/res26/cnn-* -> logits, behind the reference's plugin surface
(``register_cls`` / ``find_cls`` keys ``model.ResNet``, ``model.CNN``, ``data_loader.AudioDataLoader``,
``metric.*``, ``loss_fn.*``) and evaluator entry point (``python -m honk2_amd.run.test --config ...``).
All arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI in ``include/kws.h``
(``honk2_amd/libkws_hip.so``); PyTorch supplies device memory, streams and ``torch.distributed`` only.
"""
from . import utils, model, data_loader, metric, loss_function, dataset  # noqa: F401  (registration side effects)
from .utils import find_cls, register_cls, install_into  # noqa: F401

__version__ = "0.1.0"
