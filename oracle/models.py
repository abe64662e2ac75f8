"""CPU restatement of the reference model forward passes (test infrastructure, see oracle/__init__.py).

Two engines over the same state-dict:

* ``forward_numpy``  - plain numpy (im2col + matmul), float64 by default: the independent restatement.
* ``forward_torch``  - ``torch.nn.functional`` on the CPU in fp32: the same ops the reference's
  ``nn.Module``s dispatch; used for the fp32 comparison and as the timed ``cpu_baseline`` ("port").

ResNet follows ``/root/reference/model/resnet.py:38-60``:
    unsqueeze(1) -> conv_0 (3x3, pad 1, no bias) -> ReLU -> [AvgPool2d(pool)] -> prev = x
    for i in 1..n: x = ReLU(conv_i(x)); if i even: x = x + prev; prev = x;  x = bn_i(x)   (affine=False, eps 1e-5)
    mean over H*W -> Linear(C, n_labels)
  conv_i: dilation = padding = 2**((i-1)//3) when use_dilation else 1 (resnet.py:20-26).
CNN follows ``/root/reference/model/cnn.py:79-107``:
    unsqueeze(1) -> conv_0 (+bias, valid, stride) -> ReLU -> dropout(identity in eval) -> MaxPool(pool_0)
    [-> conv_1 -> ReLU -> MaxPool(pool_1)] -> flatten (C-major) -> [lin_0] -> [dnn_0] -> [dnn_1] -> lin_1
  with NO non-linearity between the linear layers.
"""
import numpy as np

BN_EPS = 1e-5


# ----------------------------------------------------------------------------- numpy engine
def _conv2d_np(x, w, b=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
    bsz, cin, h, wd = x.shape
    cout, _, kh, kw = w.shape
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    ho = (h + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    wo = (wd + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    iy = sh * np.arange(ho)[:, None] + dh * np.arange(kh)[None, :]      # (ho, kh)
    ix = sw * np.arange(wo)[:, None] + dw * np.arange(kw)[None, :]      # (wo, kw)
    cols = xp[:, :, iy[:, None, :, None], ix[None, :, None, :]]          # (B, C, ho, wo, kh, kw)
    cols = cols.transpose(0, 2, 3, 1, 4, 5).reshape(bsz, ho * wo, cin * kh * kw)
    out = cols @ w.reshape(cout, -1).T                                   # (B, ho*wo, cout)
    if b is not None:
        out = out + b
    return out.transpose(0, 2, 1).reshape(bsz, cout, ho, wo)


def _pool_np(x, k, op):
    kh, kw = k
    b, c, h, w = x.shape
    ho, wo = h // kh, w // kw           # floor mode, stride == kernel
    v = x[:, :, :ho * kh, :wo * kw].reshape(b, c, ho, kh, wo, kw)
    return op(op(v, axis=5), axis=3)


def _f(sd, key, dt):
    return np.asarray(sd[key], dtype=dt)


def resnet_forward_numpy(cfg, sd, feats, dtype=np.float64, taps=None):
    x = np.asarray(feats, dtype=dtype)[:, None, :, :]
    x = np.maximum(_conv2d_np(x, _f(sd, "layers.conv_0.weight", dtype), padding=(1, 1)), 0)
    if "pool" in cfg:
        x = _pool_np(x, tuple(cfg["pool"]), np.mean)
    if taps is not None:
        taps["post_pool"] = x.copy()
    prev = x
    for i in range(1, cfg["n_layers"] + 1):
        d = int(2 ** ((i - 1) // 3)) if cfg["use_dilation"] else 1
        x = np.maximum(_conv2d_np(x, _f(sd, f"layers.conv_{i}.weight", dtype), padding=(d, d), dilation=(d, d)), 0)
        if i % 2 == 0:
            x = x + prev
            prev = x
        mu = _f(sd, f"layers.bn_{i}.running_mean", dtype)[None, :, None, None]
        var = _f(sd, f"layers.bn_{i}.running_var", dtype)[None, :, None, None]
        x = (x - mu) / np.sqrt(var + dtype(BN_EPS))
        if taps is not None and i == 2:
            taps["post_layer2"] = x.copy()
    if taps is not None:
        taps["pre_mean"] = x.copy()
    m = x.reshape(x.shape[0], x.shape[1], -1).mean(axis=2)
    return m @ _f(sd, "layers.output.weight", dtype).T + _f(sd, "layers.output.bias", dtype)


def cnn_forward_numpy(cfg, sd, feats, dtype=np.float64, taps=None):
    x = np.asarray(feats, dtype=dtype)[:, None, :, :]
    for i in (0, 1):
        key = f"conv_{i}"
        if key not in cfg:
            break
        x = _conv2d_np(x, _f(sd, f"layers.{key}.weight", dtype), _f(sd, f"layers.{key}.bias", dtype)[None, None, :],
                       stride=tuple(cfg[key]["stride"]))
        x = np.maximum(x, 0)
        x = _pool_np(x, tuple(cfg[f"pool_{i}"]["kernel_size"]), np.max)
        if taps is not None:
            taps[f"post_pool_{i}"] = x.copy()
    x = x.reshape(x.shape[0], -1)
    for key in ("lin_0", "dnn_0", "dnn_1", "lin_1"):
        if key in cfg or key == "lin_1":
            x = x @ _f(sd, f"layers.{key}.weight", dtype).T + _f(sd, f"layers.{key}.bias", dtype)
    return x


def forward_numpy(model_name, cfg, sd, feats, dtype=np.float64, taps=None):
    fn = {"ResNet": resnet_forward_numpy, "CNN": cnn_forward_numpy}[model_name]
    return fn(cfg, sd, feats, dtype, taps)


# ----------------------------------------------------------------------------- torch-CPU engine
def _t(sd, key):
    import torch
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def forward_torch(model_name, cfg, sd, feats, timer=None):
    """fp32 on the CPU with torch.nn.functional; feats: (B,T,F) tensor/ndarray -> (B,n_labels) tensor.
    `timer` (ResNet): a dict that receives the seconds spent per stage (bench.py's cpu_baseline breaks its time down with it)."""
    import time
    import torch
    import torch.nn.functional as F
    x = feats if isinstance(feats, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(feats))
    x = x.float().unsqueeze(1)
    t_last = [time.perf_counter()]

    def lap(stage):
        if timer is not None:
            now = time.perf_counter()
            timer[stage] = timer.get(stage, 0.0) + now - t_last[0]
            t_last[0] = now

    with torch.no_grad():
        if model_name == "ResNet":
            x = F.relu(F.conv2d(x, _t(sd, "layers.conv_0.weight"), padding=1))
            if "pool" in cfg:
                x = F.avg_pool2d(x, tuple(cfg["pool"]))
            lap("conv_0+relu+pool")
            prev = x
            for i in range(1, cfg["n_layers"] + 1):
                d = int(2 ** ((i - 1) // 3)) if cfg["use_dilation"] else 1
                x = F.relu(F.conv2d(x, _t(sd, f"layers.conv_{i}.weight"), padding=d, dilation=d))
                if i % 2 == 0:
                    x = x + prev
                    prev = x
                x = F.batch_norm(x, _t(sd, f"layers.bn_{i}.running_mean"), _t(sd, f"layers.bn_{i}.running_var"),
                                 training=False, eps=BN_EPS)
                lap(f"conv_{i}+relu+bn")
            x = x.reshape(x.size(0), x.size(1), -1).mean(2)
            y = F.linear(x, _t(sd, "layers.output.weight"), _t(sd, "layers.output.bias"))
            lap("mean+linear")
            return y
        if model_name == "CNN":
            for i in (0, 1):
                key = f"conv_{i}"
                if key not in cfg:
                    break
                x = F.relu(F.conv2d(x, _t(sd, f"layers.{key}.weight"), _t(sd, f"layers.{key}.bias"),
                                    stride=tuple(cfg[key]["stride"])))
                x = F.max_pool2d(x, tuple(cfg[f"pool_{i}"]["kernel_size"]))
            x = x.reshape(x.size(0), -1)
            for key in ("lin_0", "dnn_0", "dnn_1", "lin_1"):
                if key in cfg or key == "lin_1":
                    x = F.linear(x, _t(sd, f"layers.{key}.weight"), _t(sd, f"layers.{key}.bias"))
            return x
    raise ValueError(model_name)


# ----------------------------------------------------------------------------- evaluation tail
def ce_loss_numpy(logits, target):
    """nn.CrossEntropyLoss()(output, target) with mean reduction (loss_function.py:6-9)."""
    z = np.asarray(logits, dtype=np.float64)
    z = z - z.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1))
    return float(np.mean(lse - z[np.arange(len(target)), np.asarray(target)]))


def accuracy_counts(logits, target, n_labels):
    """argmax (first max wins, as torch.argmax) vs target: (correct, total, per-class correct, per-class total)
    -- metric/acc.py:14-24, metric/per_class_acc.py:14-45."""
    pred = np.argmax(np.asarray(logits), axis=1)
    target = np.asarray(target)
    hit = pred == target
    pc = np.bincount(target[hit], minlength=n_labels)
    pt = np.bincount(target, minlength=n_labels)
    return int(hit.sum()), int(len(target)), pc.astype(np.int64), pt.astype(np.int64)
