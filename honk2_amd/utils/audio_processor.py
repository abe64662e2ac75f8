"""MFCC front end on the GPU (replaces reference ``utils/audio_processor.py:8-30``).

``compute_mfccs`` keeps the reference's per-clip signature (``np[n] -> np.float32 (T, n_mels, 1)``);
``compute_mfccs_batch`` is the batched device call the data loader uses.  Both run the hand-written HIP kernel
behind ``kws_mfcc``; there is no CPU implementation here.

Differences from the reference, on purpose:
* ``n_dct_filters`` is accepted and ignored exactly like the reference ignores it (``audio_processor.py:8``);
  the "DCT" the reference applies is a length-1 DCT-II == multiplication by 2, which the kernel applies.
* PCEN (``audio_processor.py:32-35``) is out of scope: dead code in the reference
  (``data_loader/audio_data_loader.py:31`` uses numpy without importing it) and needs the un-vendored
  ``pytorch-pcen`` package.
"""
import numpy as np

from .. import _lib


class AudioProcessor(object):
    def __init__(self, sr=16000, n_dct_filters=40, n_mels=40, f_max=4000, f_min=20, n_fft=480, hop_ms=10):
        self.n_mels = n_mels
        self.sr = sr
        self.f_max = f_max if f_max is not None else sr // 2
        self.f_min = f_min
        self.n_fft = n_fft
        self.hop_length = sr // 1000 * hop_ms
        self._engines = {}        # one front-end handle per GPU, created where the first tensor from that GPU arrives

    def _get_engine(self, like=None):
        import torch
        if isinstance(like, torch.Tensor) and like.is_cuda:
            index = like.device.index
        else:
            index = torch.cuda.current_device() if torch.cuda.is_available() else 0
        if index not in self._engines:
            desc = _lib.make_desc(_lib.KWS_MODEL_NONE, frontend=dict(
                sample_rate=self.sr, n_fft=self.n_fft, hop_length=self.hop_length, n_mels=self.n_mels,
                f_min=float(self.f_min), f_max=float(self.f_max)))
            self._engines[index] = _lib.Engine(desc, f"cuda:{index}" if torch.cuda.is_available() else None)
        return self._engines[index]

    def num_frames(self, n_samples):
        return 1 + n_samples // self.hop_length

    def compute_mfccs_batch(self, wav, noise=None, noise_pct=0.0):
        """(B, n) float32 or int16-PCM tensor on the GPU -> (B, T, n_mels) float32 tensor on the GPU."""
        return self._get_engine(wav).mfcc(wav, noise, noise_pct)

    def compute_mfccs_windows(self, stream, window, shift, first=0, count=None):
        """1-D float32 stream on the GPU -> features (count, T, n_mels) of the windows ``stream[i*shift : i*shift + window]``,
        ``i = first .. first + count - 1`` (the items of the reference's ``StreamingDataset``), read in place."""
        return self._get_engine(stream).mfcc_windows(stream, window, shift, first, count)

    def compute_mfccs(self, data):
        import torch
        eng = self._get_engine()
        wav = torch.as_tensor(np.asarray(data), dtype=torch.float32).reshape(1, -1).to(eng.device)
        out = eng.mfcc(wav)[0].cpu().numpy()
        return np.asfortranarray(out[:, :, None]).astype(np.float32)

    def compute_pcen(self, data):
        raise NotImplementedError("PCEN is out of scope for honk2_amd (dead code in the reference, needs pytorch-pcen)")
