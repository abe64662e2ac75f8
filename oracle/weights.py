"""Deterministic synthetic weights and inputs (test infrastructure, see oracle/__init__.py).

Weights are a pure function of (model config, seed) so the golden fixtures only
need to store seeds and expected outputs, not megabytes of parameters.  Names
and shapes are the reference's state-dict keys:

* ResNet (``model/resnet.py:11-36``): ``layers.conv_0.weight`` (C,1,3,3),
  ``layers.conv_{i}.weight`` (C,C,3,3), ``layers.bn_{i}.running_mean|running_var``
  (C), ``layers.bn_{i}.num_batches_tracked``, ``layers.output.weight|bias``.
* CNN (``model/cnn.py:12-77``): ``layers.conv_{0,1}.weight|bias``,
  ``layers.{lin_0,dnn_0,dnn_1,lin_1}.weight|bias``.

Distributions mimic PyTorch's default init (uniform +-1/sqrt(fan_in)); BN
running stats are randomised (mean ~ N(0.3, 0.2), var ~ U(0.25, 0.75)) because
fresh BatchNorm stats (0 / 1) would not exercise the normalisation
(SURVEY.md section 8c).
"""
import math
import zlib
from collections import OrderedDict

import numpy as np


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), zlib.crc32(name.encode())])))


def _uniform(seed, name, shape, bound):
    u = _rng(seed, name).random(int(np.prod(shape)), dtype=np.float64)
    return ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape)


def _normal(seed, name, shape, mean, std):
    g = _rng(seed, name)
    n = int(np.prod(shape))
    u1 = 1.0 - g.random(n, dtype=np.float64)
    u2 = g.random(n, dtype=np.float64)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def conv_out(n, k, s=1, p=0, d=1):
    return (n + 2 * p - (d * (k - 1) + 1)) // s + 1


def cnn_flatten_size(cfg):
    """Feature count after the conv/pool stack (model/cnn.py:20-46 + utils/torch_utils.py:29-65)."""
    h, w = cfg["time"], cfg["frequency"]
    c = 1
    for i in (0, 1):
        key = f"conv_{i}"
        if key not in cfg:
            break
        kh, kw = cfg[key]["kernel_size"]
        sh, sw = cfg[key]["stride"]
        c = cfg[key]["out_channels"]
        h, w = conv_out(h, kh, sh), conv_out(w, kw, sw)
        ph, pw = cfg[f"pool_{i}"]["kernel_size"]
        h, w = conv_out(h, ph, ph), conv_out(w, pw, pw)
    return c * h * w


def make_state_dict(model_name, cfg, seed=0):
    """OrderedDict name -> ndarray with the reference's key names for model `model_name`."""
    sd = OrderedDict()
    n_labels = cfg["n_labels"]
    if model_name == "ResNet":
        c = cfg["n_feature_maps"]
        sd["layers.conv_0.weight"] = _uniform(seed, "conv_0.w", (c, 1, 3, 3), 1.0 / 3.0)
        for i in range(1, cfg["n_layers"] + 1):
            sd[f"layers.conv_{i}.weight"] = _uniform(seed, f"conv_{i}.w", (c, c, 3, 3), 1.0 / math.sqrt(9 * c))
            sd[f"layers.bn_{i}.running_mean"] = _normal(seed, f"bn_{i}.m", (c,), 0.3, 0.2)
            sd[f"layers.bn_{i}.running_var"] = (0.25 + 0.5 * _rng(seed, f"bn_{i}.v").random(c)).astype(np.float32)
            sd[f"layers.bn_{i}.num_batches_tracked"] = np.asarray(1, dtype=np.int64)
        sd["layers.output.weight"] = _uniform(seed, "output.w", (n_labels, c), 1.0 / math.sqrt(c))
        sd["layers.output.bias"] = _uniform(seed, "output.b", (n_labels,), 1.0 / math.sqrt(c))
        return sd
    if model_name == "CNN":
        cin = 1
        for i in (0, 1):
            key = f"conv_{i}"
            if key not in cfg:
                break
            kh, kw = cfg[key]["kernel_size"]
            co = cfg[key]["out_channels"]
            bound = 1.0 / math.sqrt(cin * kh * kw)
            sd[f"layers.{key}.weight"] = _uniform(seed, f"{key}.w", (co, cin, kh, kw), bound)
            sd[f"layers.{key}.bias"] = _uniform(seed, f"{key}.b", (co,), bound)
            cin = co
        feat = cnn_flatten_size(cfg)
        for key in ("lin_0", "dnn_0", "dnn_1"):
            if key in cfg:
                out = cfg[key]["out_features"]
                # scaled up a little so logits of the activation-free linear chain stay O(1)
                bound = 1.0 / math.sqrt(feat)
                sd[f"layers.{key}.weight"] = _uniform(seed, f"{key}.w", (out, feat), bound)
                sd[f"layers.{key}.bias"] = _uniform(seed, f"{key}.b", (out,), bound)
                feat = out
        bound = 1.0 / math.sqrt(feat)
        sd["layers.lin_1.weight"] = _uniform(seed, "lin_1.w", (n_labels, feat), bound)
        sd["layers.lin_1.bias"] = _uniform(seed, "lin_1.b", (n_labels,), bound)
        return sd
    raise ValueError(f"unknown model {model_name}")


def make_features(batch, seed=0, time=101, freq=40):
    """(B, T, F) float32 inputs ~ N(0.65, 2.5) (the measured feature distribution); clip 0 is all-zero
    (what the reference's silence class produces, dataset/gsc_dataset.py:165-166 + audio_processor.py:27)."""
    x = _normal(seed, "features", (batch, time, freq), 0.65, 2.5)
    x[0] = 0.0
    return x


def make_waveforms(batch, n_samples=16000, seed=1234, sr=16000):
    """(B, n) float32 synthetic clips: 0.1*N(0,1) clamped to [-1,1];
    every 12th clip exact zeros; every 12th+1 a 0.5-amplitude 1 kHz sine carrying -60 dB dither
    (a bare bin-centred sine makes every off-peak mel band pure rounding noise, which no two
    fp32 implementations agree on; the undithered sine is a front-end known-answer test instead)."""
    x = np.clip(0.1 * _normal(seed, "wav", (batch, n_samples), 0.0, 1.0), -1.0, 1.0).astype(np.float32)
    t = np.arange(n_samples, dtype=np.float64) / sr
    tone = 0.5 * np.sin(2.0 * np.pi * 1000.0 * t)
    for b in range(batch):
        if b % 12 == 0:
            x[b] = 0.0
        elif b % 12 == 1:
            x[b] = (tone + 0.005 * x[b].astype(np.float64)).astype(np.float32)
    return x


def make_labels(batch, n_labels=12, seed=1234):
    return (_rng(seed, "labels").random(batch) * n_labels).astype(np.int64)


def randomize_bn_inplace(sd, seed=1):
    """Replace BN running stats of an existing state dict (torch tensors or ndarrays)."""
    for k in list(sd.keys()):
        if k.endswith("running_mean"):
            v = _normal(seed, k, tuple(sd[k].shape), 0.3, 0.2)
        elif k.endswith("running_var"):
            v = (0.25 + 0.5 * _rng(seed, k).random(tuple(sd[k].shape))).astype(np.float32)
        else:
            continue
        try:
            import torch
            if isinstance(sd[k], torch.Tensor):
                sd[k] = torch.from_numpy(v)
                continue
        except ImportError:
            pass
        sd[k] = v
    return sd
