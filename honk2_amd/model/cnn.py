"""The Sainath-Parada ``cnn-*`` family on MI355X.

Plugin key and constructor contract of the reference (``model/cnn.py:10-77``): ``find_cls("model.CNN")``,
``CNN(config)`` with ``time, frequency, dropout_prob, conv_0{out_channels,kernel_size,stride}, pool_0{kernel_size}``
and optional ``conv_1/pool_1, lin_0, dnn_0, dnn_1``; the flatten size is derived with the same arithmetic
(``utils/torch_utils.py:29-65``).  ``forward`` (reference ``:79-107``: conv+bias -> ReLU -> MaxPool, flatten,
Linear chain WITHOUT non-linearities, dropout = identity in eval) is one call into libkws_hip.so.
"""
import numpy as np
import torch.nn as nn

from .. import _lib
from ..utils import register_cls, calculate_conv_output_size, calculate_pool_output_size
from .model_utils import BaseModel, ConvParams, LinearParams


@register_cls('model.CNN')
class CNN(BaseModel):
    def __init__(self, config):
        super().__init__()
        self.config = dict(config)
        self.layers = nn.ModuleDict()
        size = [1, config["time"], config["frequency"]]
        for i in (0, 1):
            key = f"conv_{i}"
            if key not in config:
                break
            spec = config[key]
            self.layers[key] = ConvParams(size[0], spec["out_channels"], tuple(spec["kernel_size"]), bias=True)
            size = [spec["out_channels"]] + calculate_conv_output_size(size[1:], spec["kernel_size"], stride=spec["stride"])
            size = [size[0]] + calculate_pool_output_size(size[1:], config[f"pool_{i}"]["kernel_size"])
        features = int(np.prod(size))
        self.flatten_size = features
        for key in ("lin_0", "dnn_0", "dnn_1"):
            if key in config:
                self.layers[key] = LinearParams(features, config[key]["out_features"])
                features = config[key]["out_features"]
        self.layers["lin_1"] = LinearParams(features, config["n_labels"])
        self.dropout_prob = config["dropout_prob"]

    def _make_desc(self):
        c = self.config
        d = _lib.make_desc(_lib.KWS_MODEL_CNN, n_labels=c["n_labels"], time=c["time"], freq=c["frequency"],
                           dtype=c.get("dtype", "f32"))
        n = 0
        for i in (0, 1):
            if f"conv_{i}" not in c:
                break
            spec = c[f"conv_{i}"]
            d.conv[i].out_channels = spec["out_channels"]
            d.conv[i].kernel_h, d.conv[i].kernel_w = spec["kernel_size"]
            d.conv[i].stride_h, d.conv[i].stride_w = spec["stride"]
            d.pool_kh[i], d.pool_kw[i] = c[f"pool_{i}"]["kernel_size"]
            n += 1
        d.n_conv = n
        d.lin0_out = c["lin_0"]["out_features"] if "lin_0" in c else 0
        d.dnn0_out = c["dnn_0"]["out_features"] if "dnn_0" in c else 0
        d.dnn1_out = c["dnn_1"]["out_features"] if "dnn_1" in c else 0
        return d
