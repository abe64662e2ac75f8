"""res8 / res15 / res26 (+ narrow) on MI355X.

Plugin key and constructor contract of the reference (``model/resnet.py:9-36``): ``find_cls("model.ResNet")``,
``ResNet(config)`` with ``n_layers, n_feature_maps, use_dilation, n_labels`` and an optional ``pool`` key
(any other spelling, e.g. hey_snips' ``avg_pool``, is ignored exactly as the reference ignores it,
``model/resnet.py:29``).  ``forward`` (reference ``:38-60``) is a single call into libkws_hip.so: the fused
res8 kernel when the config matches ``config/resnet/res8.json``, the layer-wise MFMA kernels otherwise.
"""
import torch.nn as nn

from .. import _lib
from ..utils import register_cls
from .model_utils import BaseModel, BatchNormStats, ConvParams, LinearParams


@register_cls('model.ResNet')
class ResNet(BaseModel):
    def __init__(self, config):
        super().__init__()
        self.config = dict(config)
        self.n_layers = config["n_layers"]
        n_maps = config["n_feature_maps"]
        self.layers = nn.ModuleDict()
        self.layers["conv_0"] = ConvParams(1, n_maps, (3, 3), bias=False)
        for i in range(1, self.n_layers + 1):
            self.layers[f"conv_{i}"] = ConvParams(n_maps, n_maps, (3, 3), bias=False)
            self.layers[f"bn_{i}"] = BatchNormStats(n_maps)
        self.layers["output"] = LinearParams(n_maps, config["n_labels"])

    def dilation(self, i):
        return int(2 ** ((i - 1) // 3)) if self.config["use_dilation"] else 1

    def _make_desc(self):
        c = self.config
        pool = c.get("pool", (0, 0))
        return _lib.make_desc(_lib.KWS_MODEL_RESNET, n_labels=c["n_labels"], n_layers=c["n_layers"],
                              n_feature_maps=c["n_feature_maps"], use_dilation=int(bool(c["use_dilation"])),
                              pool_h=int(pool[0]), pool_w=int(pool[1]), dtype=c.get("dtype", "f32"))
