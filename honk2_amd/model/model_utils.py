"""Model base class and parameter containers.

``BaseModel`` keeps the reference's API (``model/model_utils.py:6-11``: ``num_params``,
``num_trainable_params``) on top of ``torch.nn.Module`` so ``.eval()``, ``.to()``, ``.state_dict()`` and
``.load_state_dict()`` behave as in the reference.  The modules below only HOLD tensors under the reference's
state-dict key names (``layers.conv_3.weight``, ``layers.bn_3.running_var`` ...); they have no forward of their
own, because the whole forward pass is one call into the HIP library (``kws_forward``).
Initialisation draws from the torch RNG in the same order and with the same distributions as the
``nn.Conv2d`` / ``nn.Linear`` modules the reference builds, so ``torch.manual_seed(s); Model(cfg)`` yields the
same weights as the reference.
"""
import ctypes as C
import math
import threading
import weakref
from abc import ABC

import torch
import torch.nn as nn

from .. import _lib


class ConvParams(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, bias):
        super().__init__()
        kh, kw = kernel_size
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kh, kw))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(in_channels * kh * kw)
            self.bias = nn.Parameter(torch.empty(out_channels))
            nn.init.uniform_(self.bias, -bound, bound)


class LinearParams(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_features)
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.uniform_(self.bias, -bound, bound)


class BatchNormStats(nn.Module):
    """Running statistics of ``nn.BatchNorm2d(C, affine=False)`` (eval mode only, eps = 1e-5)."""

    def __init__(self, channels):
        super().__init__()
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class BaseModel(ABC, nn.Module):
    def __init__(self):
        super().__init__()
        # One engine (kws_handle + weights + workspace) per GPU the module's tensors have been seen on.  The dict and its
        # lock are shared -- by reference -- between the shallow per-device replicas ``nn.DataParallel`` makes (the
        # reference's multi-GPU mechanism, ``run/test.py:69-70``): each replica thread finds or creates the engine of its
        # own device and never touches (let alone destroys) another replica's handle.
        self._engines = {}
        self._engines_lock = threading.Lock()
        # torch.nn.parallel.replicate() builds a replica as a shallow copy of __dict__ with an EMPTY _parameters dict (the broadcast
        # copies hang on it as plain attributes and in _former_parameters): a replica has no parameters() and its state_dict() lacks
        # every weight.  This entry is copied with the rest of __dict__, so a replica can find the module it was made from, whose
        # tensors are what every device's engine is loaded from (engine()).
        object.__setattr__(self, "_origin", weakref.ref(self))

    # engines are per-process device state: a deep copy / pickle of the module carries the tensors only and builds its own
    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_engines", None)
        state.pop("_engines_lock", None)
        state.pop("_origin", None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._engines = {}
        self._engines_lock = threading.Lock()
        object.__setattr__(self, "_origin", weakref.ref(self))

    def num_params(self):
        return sum(p.numel() for p in self.parameters())

    def num_trainable_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    # ------------------------------------------------------------------ HIP engine plumbing
    def _make_desc(self):
        raise NotImplementedError

    def _source(self):
        """The module whose tensors define the weights: this one, or -- for a DataParallel replica -- the module it was replicated
        from (a replica's own tensors are per-forward broadcast copies that may reuse an address at version 0)."""
        src = self._origin() if getattr(self, "_is_replica", False) else self
        return self if src is None else src

    def _weights_key(self):
        return tuple((k, v.data_ptr(), v._version) for k, v in self._source().state_dict(keep_vars=True).items())

    def _own_device(self):
        """Device of this module's own tensors, replicas included (their parameters sit in _former_parameters); None if it has none."""
        for m in self.modules():
            for holder in (getattr(m, "_former_parameters", None), m._parameters, m._buffers):
                for t in (holder or {}).values():
                    if t is not None:
                        return t.device
        return None

    def engine(self, device=None):
        """The kws_handle of the device this module's tensors live on (or of `device`: the input's, in forward), with the current
        weights loaded (re-uploaded when they change).  Engines of other devices stay alive until the module is garbage-collected."""
        if device is None:
            device = self._own_device()                   # (tensors still on the host: the current GPU computes)
        if not torch.cuda.is_available():
            raise RuntimeError("honk2_amd: no ROCm device visible to PyTorch; the HIP path is mandatory (no CPU fallback)")
        device = torch.device("cpu") if device is None else torch.device(device)
        index = device.index if device.type == "cuda" and device.index is not None else torch.cuda.current_device()
        key = self._weights_key()
        with self._engines_lock:
            slot = self._engines.get(index)
            if slot is None:
                slot = self._engines[index] = [_lib.Engine(self._make_desc(), torch.device("cuda", index)), None]
            if slot[1] != key:
                for name, tensor in self._source().state_dict().items():
                    slot[0].load_tensor(name, tensor)
                slot[1] = key
            return slot[0]

    def _require_eval(self):
        if self.training:
            raise RuntimeError("honk2_amd models are inference-only: call model.eval() first "
                               "(BatchNorm uses running statistics, dropout is the identity)")

    def forward(self, x, out=None):
        """(B, T, F) float32 features on the GPU -> (B, n_labels) logits (one ``kws_forward`` call); `out`: write them into this tensor."""
        self._require_eval()
        with torch.no_grad():
            return self.engine().forward(x, out)

    def forward_wav(self, wav, out=None, noise=None, noise_pct=0.0):
        """(B, n_samples) waveforms on the GPU -> logits, front end fused in.  float32 input -> ``kws_forward_wav``;
        int16 PCM input -> ``kws_forward_pcm16`` (x/32768, optional ``+ noise * noise_pct`` as
        ``GSCDataset.__getitem__`` does, both inside the front end's staging load)."""
        self._require_eval()
        with torch.no_grad():
            return self.engine().forward_wav(wav, out, noise, noise_pct)

    def forward_windows(self, stream, window, shift, first=0, count=None, out=None):
        """Streaming evaluation: 1-D float32 stream on the GPU -> logits of the windows ``stream[i*shift : i*shift +
        window]`` (reference ``dataset/dataset_utils.py:20-98``), read in place by ``kws_forward_windows``."""
        self._require_eval()
        with torch.no_grad():
            return self.engine().forward_windows(stream, window, shift, first, count, out)

    def plan_name(self):
        return self.engine().plan_name()

    def plan_detail(self):
        return self.engine().plan_detail()

    def chunk_clips(self, batch, frames=101):
        return self.engine().chunk_clips(batch, frames)
