"""Read-only compatibility with checkpoints written by the reference's ``Workspace._save``
(``utils/workspace.py:34-45``): a dict holding ``model_state_dict`` plus pickled loss/metric objects and
optimizer state.  Only the model weights matter for inference.

The reference's own ``Workspace._load`` (``utils/workspace.py:58-70``) calls ``torch.load`` without
``weights_only=False`` and fails on torch >= 2.6 because the checkpoint pickles ``loss_function.ce_loss`` and
``metric.acc.Acc`` by module reference; here those names are mapped onto this package's equivalents.
"""
import pickle

import torch


class _Unpickler(pickle.Unpickler):
    _REMAP = {"loss_function": "honk2_amd.loss_function", "metric": "honk2_amd.metric",
              "metric.acc": "honk2_amd.metric.acc", "metric.per_class_acc": "honk2_amd.metric.per_class_acc",
              "metric.metric_utils": "honk2_amd.metric.metric_utils"}

    def find_class(self, module, name):
        return super().find_class(self._REMAP.get(module, module), name)


class _PickleModule:
    Unpickler = _Unpickler
    load = staticmethod(pickle.load)
    __name__ = "pickle"


def load_checkpoint_state(path, map_location="cpu"):
    """Return (model_state_dict with any DataParallel 'module.' prefix stripped, rest of the checkpoint dict)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_PickleModule)
    sd = ckpt.pop("model_state_dict")
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    return sd, ckpt
