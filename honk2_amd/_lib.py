"""ctypes binding of libkws_hip.so (C ABI: include/kws.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (``make -C honk2_amd/csrc``).  There is NO
CPU fallback: if the library is missing, cannot be loaded, or no HIP device is present, every compute entry
point raises ``RuntimeError``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KWS_LIB points experiments (A/B builds of one kernel, the EXPERIMENTS=1 build with its debug / timing switches) at another build of the
# same library; default = the in-tree product build
EXP_LIB_PATH = os.path.join(_HERE, "libkws_hip_exp.so")
LIB_PATH = os.environ.get("KWS_LIB") or os.path.join(_HERE, "libkws_hip.so")

KWS_MODEL_NONE, KWS_MODEL_RESNET, KWS_MODEL_CNN = 0, 1, 2
KWS_OK, KWS_EINVAL, KWS_ENOMEM, KWS_EUNSUPPORTED, KWS_ENOWORKSPACE, KWS_ENOWEIGHTS, KWS_EHIP = 0, -1, -2, -3, -4, -5, -6
KWS_DTYPE_F32, KWS_DTYPE_BF16X3, KWS_DTYPE_BF16, KWS_DTYPE_F16 = 0, 1, 2, 3
DTYPES = {"f32": KWS_DTYPE_F32, "fp32": KWS_DTYPE_F32, "float32": KWS_DTYPE_F32, "bf16x3": KWS_DTYPE_BF16X3,
          "bf16": KWS_DTYPE_BF16, "bfloat16": KWS_DTYPE_BF16, "fp16": KWS_DTYPE_F16, "f16": KWS_DTYPE_F16,
          "float16": KWS_DTYPE_F16}

# every symbol include/kws.h declares (tests check the library exports exactly these)
EXPORTS = (
    "kws_create", "kws_destroy", "kws_load_weights", "kws_workspace_bytes", "kws_chunk_clips", "kws_set_workspace", "kws_num_frames",
    "kws_mfcc", "kws_mfcc_pcm16", "kws_forward", "kws_forward_wav", "kws_forward_pcm16", "kws_workspace_bytes_windows", "kws_mfcc_windows", "kws_forward_windows", "kws_eval_batch", "kws_plan_name", "kws_plan_detail", "kws_profile_enable",
    "kws_profile_read", "kws_last_error", "kws_abi_version",
)


class ConvDesc(C.Structure):
    _fields_ = [("out_channels", C.c_int32), ("kernel_h", C.c_int32), ("kernel_w", C.c_int32),
                ("stride_h", C.c_int32), ("stride_w", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("family", C.c_int32), ("dtype", C.c_int32), ("n_labels", C.c_int32),
        ("time", C.c_int32), ("freq", C.c_int32),
        ("n_layers", C.c_int32), ("n_feature_maps", C.c_int32), ("use_dilation", C.c_int32),
        ("pool_h", C.c_int32), ("pool_w", C.c_int32),
        ("n_conv", C.c_int32), ("conv", ConvDesc * 2), ("pool_kh", C.c_int32 * 2), ("pool_kw", C.c_int32 * 2),
        ("lin0_out", C.c_int32), ("dnn0_out", C.c_int32), ("dnn1_out", C.c_int32),
        ("sample_rate", C.c_int32), ("n_fft", C.c_int32), ("hop_length", C.c_int32), ("n_mels", C.c_int32),
        ("f_min", C.c_float), ("f_max", C.c_float),
    ]


_lib = None


def load():
    """Load libkws_hip.so once; raises RuntimeError (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"honk2_amd: HIP extension {LIB_PATH} is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C honk2_amd/csrc`). "
            "There is no CPU fallback.")
    _lib = bind(LIB_PATH)
    return _lib


def bind(path):
    """dlopen one build of the library and declare the C signatures of include/kws.h on it (`load()` for the product; tests bind the
    experiments build -- `make -C honk2_amd/csrc EXPERIMENTS=1` -- beside it)."""
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise RuntimeError(f"honk2_amd: cannot load {path}: {e}") from e
    vp, ci, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.kws_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(vp)]
    lib.kws_create.restype = ci
    lib.kws_destroy.argtypes = [vp]
    lib.kws_destroy.restype = None
    lib.kws_load_weights.argtypes = [vp, C.c_char_p, vp, sz]
    lib.kws_load_weights.restype = ci
    lib.kws_workspace_bytes.argtypes = [vp, ci, ci]
    lib.kws_workspace_bytes.restype = sz
    lib.kws_chunk_clips.argtypes = [vp, ci, ci]
    lib.kws_chunk_clips.restype = ci
    lib.kws_set_workspace.argtypes = [vp, vp, sz]
    lib.kws_set_workspace.restype = ci
    lib.kws_num_frames.argtypes = [vp, ci]
    lib.kws_num_frames.restype = ci
    lib.kws_mfcc.argtypes = [vp, vp, ci, ci, vp, vp]
    lib.kws_mfcc.restype = ci
    lib.kws_mfcc_pcm16.argtypes = [vp, vp, vp, C.c_float, ci, ci, vp, vp]
    lib.kws_mfcc_pcm16.restype = ci
    lib.kws_forward_pcm16.argtypes = [vp, vp, vp, C.c_float, ci, ci, vp, vp]
    lib.kws_forward_pcm16.restype = ci
    lib.kws_forward.argtypes = [vp, vp, ci, ci, vp, vp]
    lib.kws_forward.restype = ci
    lib.kws_forward_wav.argtypes = [vp, vp, ci, ci, vp, vp]
    lib.kws_forward_wav.restype = ci
    lib.kws_workspace_bytes_windows.argtypes = [vp, ci, ci, ci]
    lib.kws_workspace_bytes_windows.restype = sz
    lib.kws_mfcc_windows.argtypes = [vp, vp, sz, ci, ci, ci, vp, vp]
    lib.kws_mfcc_windows.restype = ci
    lib.kws_forward_windows.argtypes = [vp, vp, sz, ci, ci, ci, vp, vp]
    lib.kws_forward_windows.restype = ci
    lib.kws_eval_batch.argtypes = [vp, vp, vp, ci, vp, vp, vp]
    lib.kws_eval_batch.restype = ci
    lib.kws_plan_name.argtypes = [vp]
    lib.kws_plan_name.restype = C.c_char_p
    lib.kws_plan_detail.argtypes = [vp]
    lib.kws_plan_detail.restype = C.c_char_p
    lib.kws_profile_enable.argtypes = [vp, ci]
    lib.kws_profile_enable.restype = ci
    lib.kws_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(ci)]
    lib.kws_profile_read.restype = ci
    lib.kws_last_error.argtypes = []
    lib.kws_last_error.restype = C.c_char_p
    lib.kws_abi_version.argtypes = []
    lib.kws_abi_version.restype = ci
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().kws_last_error().decode(errors="replace")
        raise RuntimeError(f"honk2_amd: {what} failed ({rc}): {msg}")


FRONTEND_DEFAULTS = dict(sample_rate=16000, n_fft=480, hop_length=160, n_mels=40, f_min=20.0, f_max=4000.0)


def make_desc(family, n_labels=0, frontend=None, **kw):
    d = ModelDesc()
    d.struct_size = C.sizeof(ModelDesc)
    d.family = family
    d.dtype = DTYPES[str(kw.pop("dtype", "f32")).lower()]
    d.n_labels = n_labels
    fe = dict(FRONTEND_DEFAULTS)
    fe.update(frontend or {})
    for k, v in fe.items():
        setattr(d, k, v)
    d.time = kw.pop("time", 101)
    d.freq = kw.pop("freq", fe["n_mels"])
    for k, v in kw.items():
        setattr(d, k, v)
    return d


class Engine:
    """Owns one kws_handle plus its torch-allocated workspace.  Device memory for I/O and scratch comes from
    PyTorch (plumbing); all arithmetic happens inside libkws_hip.so."""

    def __init__(self, desc, device=None):
        """`device`: the GPU the handle lives on (weights, workspace, launches); default = torch's current device.
        Every tensor handed to the compute methods must be on that device."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("honk2_amd: no ROCm device visible to PyTorch; the HIP path is mandatory (no CPU fallback)")
        self.lib = load()
        self.desc = desc
        self.handle = C.c_void_p()
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("honk2_amd: an Engine needs a ROCm device; there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device() if device.index is None else device.index)
        with torch.cuda.device(self.device):               # kws_create binds the handle to the current HIP device
            check(self.lib.kws_create(C.byref(desc), C.byref(self.handle)), "kws_create")
        self._ws = None
        self._captured = False      # a compute call of this engine has been captured into a graph
        self._parked = []           # workspaces outgrown since then: captured graphs name them, so they stay alive with the engine

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.kws_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters
    def load_tensor(self, name, tensor):
        import torch
        t = tensor.detach()
        if t.dtype == torch.int64:     # num_batches_tracked
            return
        host = t.to(device="cpu", dtype=torch.float32).contiguous()
        check(self.lib.kws_load_weights(self.handle, name.encode(), C.c_void_p(host.data_ptr()),
                                        host.numel() * 4), f"kws_load_weights({name})")

    # ---- workspace
    def _ensure_ws(self, batch, frames):
        self._ensure_ws_bytes(int(self.lib.kws_workspace_bytes(self.handle, batch, frames)))

    def _ensure_ws_bytes(self, need):
        """The workspace grows with the largest call seen.  Buffers baked into a captured graph must outlive every replay
        (include/kws.h): once a call of this engine has been captured, an outgrown workspace is kept instead of being handed
        back to the caching allocator, and a call that would have to grow it DURING capture is refused."""
        import torch
        capturing = torch.cuda.is_current_stream_capturing()
        self._captured = self._captured or capturing
        if self._ws is None or self._ws.numel() < need:
            if capturing:
                raise RuntimeError("honk2_amd: this call needs a larger workspace and the stream is being captured -- make one un-captured "
                                   "warm-up call with this (or a larger) batch size first")
            if self._captured and self._ws is not None:
                self._parked.append(self._ws)
            self._ws = None
            self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
            check(self.lib.kws_set_workspace(self.handle, C.c_void_p(self._ws.data_ptr()), self._ws.numel()),
                  "kws_set_workspace")

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)   # the caller's stream on the handle's device

    def _on_device(self, t, what):
        if t.device != self.device:
            raise RuntimeError(f"honk2_amd: {what} is on {t.device} but this engine (weights, workspace) lives on "
                               f"{self.device}; move the model with .to(...) or the tensor")

    def _check_in(self, t, ndim, what):
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda):
            raise RuntimeError(f"honk2_amd: {what} must be a CUDA(ROCm) tensor; there is no CPU path")
        self._on_device(t, what)
        if t.dim() != ndim:
            raise ValueError(f"honk2_amd: {what} must have {ndim} dimensions, got {tuple(t.shape)}")
        return t.to(dtype=torch.float32).contiguous()

    # ---- compute
    def num_frames(self, n_samples):
        return int(self.lib.kws_num_frames(self.handle, int(n_samples)))

    def _wave_args(self, wav, noise, noise_pct):
        """-> (tensor, is_pcm16, noise tensor or None).  int16 tensors take the fused PCM path."""
        import torch
        if not (isinstance(wav, torch.Tensor) and wav.is_cuda):
            raise RuntimeError("honk2_amd: wav must be a CUDA(ROCm) tensor; there is no CPU path")
        if wav.dim() != 2:
            raise ValueError(f"honk2_amd: wav must have 2 dimensions, got {tuple(wav.shape)}")
        self._on_device(wav, "wav")
        pcm = wav.dtype == torch.int16
        wav = wav.contiguous() if pcm else wav.to(dtype=torch.float32).contiguous()
        if noise is not None:
            noise = self._check_in(noise, 2, "noise")
            if noise.shape != wav.shape:
                raise ValueError("honk2_amd: noise must have the shape of wav")
            if not pcm:                                   # float input: mix on the device with torch (plumbing)
                wav, noise = wav + noise * float(noise_pct), None
        return wav, pcm, noise

    def mfcc(self, wav, noise=None, noise_pct=0.0):
        import torch
        wav, pcm, noise = self._wave_args(wav, noise, noise_pct)
        b, n = wav.shape
        t = self.num_frames(n)
        out = torch.empty((b, t, self.desc.n_mels), dtype=torch.float32, device=wav.device)
        if b == 0:
            return out
        if pcm:
            check(self.lib.kws_mfcc_pcm16(self.handle, C.c_void_p(wav.data_ptr()),
                                          C.c_void_p(noise.data_ptr()) if noise is not None else None,
                                          C.c_float(noise_pct), b, n, C.c_void_p(out.data_ptr()), self._stream()),
                  "kws_mfcc_pcm16")
        else:
            check(self.lib.kws_mfcc(self.handle, C.c_void_p(wav.data_ptr()), b, n, C.c_void_p(out.data_ptr()),
                                    self._stream()), "kws_mfcc")
        return out

    def forward(self, feat, out=None):
        import torch
        feat = self._check_in(feat, 3, "features")
        b, t, f = feat.shape
        if f != self.desc.freq:
            raise ValueError(f"honk2_amd: expected {self.desc.freq} frequency bins, got {f}")
        if b:
            self._ensure_ws(b, t)
        if out is None:
            out = torch.empty((b, self.desc.n_labels), dtype=torch.float32, device=feat.device)
        elif tuple(out.shape) != (b, self.desc.n_labels) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != feat.device:
            raise ValueError("honk2_amd: `out` must be a contiguous float32 (B, n_labels) tensor on the input's device")
        if b == 0:
            return out
        check(self.lib.kws_forward(self.handle, C.c_void_p(feat.data_ptr()), b, t, C.c_void_p(out.data_ptr()),
                                   self._stream()), "kws_forward")
        return out

    def forward_wav(self, wav, out=None, noise=None, noise_pct=0.0):
        import torch
        wav, pcm, noise = self._wave_args(wav, noise, noise_pct)
        b, n = wav.shape
        if b:
            self._ensure_ws(b, self.num_frames(n))
        if out is None:
            out = torch.empty((b, self.desc.n_labels), dtype=torch.float32, device=wav.device)
        if b == 0:
            return out
        if pcm:
            check(self.lib.kws_forward_pcm16(self.handle, C.c_void_p(wav.data_ptr()),
                                             C.c_void_p(noise.data_ptr()) if noise is not None else None,
                                             C.c_float(noise_pct), b, n, C.c_void_p(out.data_ptr()), self._stream()),
                  "kws_forward_pcm16")
        else:
            check(self.lib.kws_forward_wav(self.handle, C.c_void_p(wav.data_ptr()), b, n,
                                           C.c_void_p(out.data_ptr()), self._stream()), "kws_forward_wav")
        return out

    # ---- streaming: overlapping windows of one long waveform, read in place (reference dataset/dataset_utils.py:20-98)
    def _stream_args(self, stream, window, shift, first, count):
        import torch
        stream = self._check_in(stream, 1, "stream")
        if stream.dtype != torch.float32:
            raise ValueError("honk2_amd: the stream must be float32")
        n = stream.numel()
        total = 0 if n < window else (n - window) // shift + 1
        if count is None:
            count = max(total - first, 0)
        if first < 0 or count < 0 or first + count > total:
            raise ValueError(f"honk2_amd: windows [{first}, {first + count}) outside the stream's {total} windows")
        if (first * shift) % 4:
            raise ValueError("honk2_amd: first * shift must be a multiple of 4 samples (16-byte aligned window start)")
        return stream, n - first * shift, first * shift * 4, count

    def mfcc_windows(self, stream, window, shift, first=0, count=None):
        import torch
        stream, n_left, byte_off, count = self._stream_args(stream, window, shift, first, count)
        out = torch.empty((count, self.num_frames(window), self.desc.n_mels), dtype=torch.float32, device=stream.device)
        if count:
            self._ensure_ws_bytes(int(self.lib.kws_workspace_bytes_windows(self.handle, window, shift, count)))
            check(self.lib.kws_mfcc_windows(self.handle, C.c_void_p(stream.data_ptr() + byte_off), n_left, window, shift,
                                            count, C.c_void_p(out.data_ptr()), self._stream()), "kws_mfcc_windows")
        return out

    def forward_windows(self, stream, window, shift, first=0, count=None, out=None):
        import torch
        stream, n_left, byte_off, count = self._stream_args(stream, window, shift, first, count)
        if count:
            self._ensure_ws_bytes(int(self.lib.kws_workspace_bytes_windows(self.handle, window, shift, count)))
        if out is None:
            out = torch.empty((count, self.desc.n_labels), dtype=torch.float32, device=stream.device)
        if count:
            check(self.lib.kws_forward_windows(self.handle, C.c_void_p(stream.data_ptr() + byte_off), n_left, window,
                                               shift, count, C.c_void_p(out.data_ptr()), self._stream()),
                  "kws_forward_windows")
        return out

    def eval_batch(self, logits, target, stats, loss_sum):
        import torch
        for t, what in ((logits, "logits"), (target, "target"), (stats, "stats"), (loss_sum, "loss_sum")):
            self._on_device(t, what)
        n = self.desc.n_labels
        if stats.dtype != torch.int64 or stats.numel() < 3 + 2 * n or not stats.is_contiguous():
            raise ValueError(f"honk2_amd: stats must be a contiguous int64 tensor of at least 3 + 2 * n_labels = {3 + 2 * n} "
                             f"words (ABI version 2: the last word counts out-of-range targets), got {stats.dtype} x {stats.numel()}")
        if loss_sum.dtype != torch.float64 or loss_sum.numel() < 1:
            raise ValueError("honk2_amd: loss_sum must be a float64 tensor")
        if target.dtype != torch.int64 or target.numel() < logits.shape[0] or logits.dtype != torch.float32 \
                or logits.dim() != 2 or logits.shape[1] != n or not logits.is_contiguous():
            raise ValueError("honk2_amd: eval_batch takes contiguous float32 (B, n_labels) logits and int64 targets")
        check(self.lib.kws_eval_batch(self.handle, C.c_void_p(logits.data_ptr()), C.c_void_p(target.data_ptr()),
                                      logits.shape[0], C.c_void_p(stats.data_ptr()), C.c_void_p(loss_sum.data_ptr()),
                                      self._stream()), "kws_eval_batch")

    def plan_name(self):
        return self.lib.kws_plan_name(self.handle).decode()

    def plan_detail(self):
        return self.lib.kws_plan_detail(self.handle).decode()

    def chunk_clips(self, batch, frames):
        """Clips per chunk of a forward call of `batch` clips (kws_chunk_clips): the unit the fp16 range guard recomputes."""
        return int(self.lib.kws_chunk_clips(self.handle, int(batch), int(frames)))

    def profile_enable(self, on=True):
        check(self.lib.kws_profile_enable(self.handle, int(bool(on))), "kws_profile_enable")

    def profile_read(self):
        m, f, n = C.c_double(), C.c_double(), C.c_int()
        check(self.lib.kws_profile_read(self.handle, C.byref(m), C.byref(f), C.byref(n)), "kws_profile_read")
        return m.value, f.value, n.value
