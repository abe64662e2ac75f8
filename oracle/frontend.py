"""CPU restatement of the reference front end (test infrastructure, see oracle/__init__.py).

Follows ``/root/reference/utils/audio_processor.py:8-30``:

    melspectrogram(y, sr=16000, n_mels=40, hop_length=160, n_fft=480, fmin=20, fmax=4000)   :19-26
    data[data > 0] = log(data[data > 0])                                                     :27
    [scipy.fftpack.dct(x) for x in split(data, T, axis=1)]  -> DCT-II over an axis of len 1  :28
    np.array(..., order="F").astype(float32)  -> shape (T, n_mels, 1)                        :29

librosa (< 0.10) semantics restated here (PARITY UNPINNED, librosa is absent):
``center=True`` reflect padding by n_fft//2, periodic Hann window of length
n_fft, frames at hop_length, rFFT, ``power=2.0``, Slaney mel filterbank
(``htk=False``, ``norm='slaney'``) cast to float32, ``mel_basis.dot(S)``.

The DCT step: ``scipy.fftpack.dct(x)`` is DCT-II, unnormalised, along the LAST
axis; every split has shape (n_mels, 1), so the transform length is 1 and the
result is exactly ``2 * x``.  (Pinned against scipy in the tests.)
"""
import numpy as np

SR = 16000
N_FFT = 480
HOP = 160
N_MELS = 40
F_MIN = 20.0
F_MAX = 4000.0


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3.0)
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_part = 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) / logstep
    return np.where(f >= 1000.0, log_part, lin)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    lin = m * (200.0 / 3.0)
    logstep = np.log(6.4) / 27.0
    return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), lin)


def mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=F_MIN, fmax=F_MAX, dtype=np.float32):
    """Slaney-normalised triangular filterbank, shape (n_mels, 1 + n_fft//2)."""
    n_bins = 1 + n_fft // 2
    fft_hz = np.linspace(0.0, sr / 2.0, n_bins)
    edges = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    widths = np.diff(edges)
    dist = edges[:, None] - fft_hz[None, :]
    bank = np.zeros((n_mels, n_bins), dtype=dtype)
    for i in range(n_mels):
        rising = -dist[i] / widths[i]
        falling = dist[i + 2] / widths[i + 1]
        bank[i] = np.maximum(0.0, np.minimum(rising, falling))
    area_norm = 2.0 / (edges[2:n_mels + 2] - edges[:n_mels])
    bank *= area_norm[:, None].astype(dtype)
    return bank


def hann_periodic(n=N_FFT, dtype=np.float64):
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(dtype)


def num_frames(n_samples, hop=HOP):
    # centre padding by n_fft//2 each side: 1 + (n + 2*(n_fft//2) - n_fft) // hop
    return 1 + n_samples // hop


def frame_signal(y, n_fft=N_FFT, hop=HOP):
    """(B, n) -> (B, T, n_fft) after reflect padding n_fft//2 (np.pad mode='reflect')."""
    y = np.atleast_2d(y)
    padded = np.pad(y, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
    t = 1 + (padded.shape[1] - n_fft) // hop
    idx = hop * np.arange(t)[:, None] + np.arange(n_fft)[None, :]
    return padded[:, idx]


def power_spectrogram(y, n_fft=N_FFT, hop=HOP, precision="f64"):
    """|STFT|^2, shape (B, T, 1+n_fft//2).  precision 'f32' mimics complex64 arithmetic."""
    if precision == "f32":
        frames = frame_signal(np.asarray(y, dtype=np.float32), n_fft, hop)
        win = hann_periodic(n_fft, np.float32)
        import scipy.fft
        spec = scipy.fft.rfft((frames * win).astype(np.float32), axis=-1)  # complex64 for f32 input
        mag = np.abs(spec).astype(np.float32)
        return mag * mag
    frames = frame_signal(np.asarray(y, dtype=np.float64), n_fft, hop)
    spec = np.fft.rfft(frames * hann_periodic(n_fft), axis=-1)
    return spec.real ** 2 + spec.imag ** 2


def mel_power(y, precision="f64", **kw):
    """(B, T, n_mels) mel power before the log."""
    sr = kw.get("sr", SR); n_fft = kw.get("n_fft", N_FFT); hop = kw.get("hop", HOP)
    n_mels = kw.get("n_mels", N_MELS); fmin = kw.get("fmin", F_MIN); fmax = kw.get("fmax", F_MAX)
    p = power_spectrogram(y, n_fft, hop, precision)
    bank = mel_filterbank(sr, n_fft, n_mels, fmin, fmax, np.float32)
    if precision == "f32":
        return np.matmul(p.astype(np.float32), bank.T.astype(np.float32))
    return np.matmul(p, bank.T.astype(np.float64))


def dct_len1(x):
    """DCT-II, unnormalised, over an axis of length 1 (what audio_processor.py:28 computes)."""
    return 2.0 * x


def compute_mfccs_batch(y, precision="f64", workers=1, **kw):
    """(B, n) waveforms -> (B, T, n_mels) float32 features, [b,t,f] = 2*ln(mel[f,t]) where mel>0 else 0.

    This is the tensor the reference's collate_fn hands to the model
    (data_loader/audio_data_loader.py:28, reshape(1,-1,40) of the (T,40,1) array).
    `workers` > 1: blocks of clips on a thread pool (numpy / scipy release the GIL in every heavy step); a clip's arithmetic
    does not depend on its block, so the result is the same array (bench.py's timed CPU leg uses it; the reference itself
    spreads clips over DataLoader worker processes, data_loader/audio_data_loader.py:10-21).
    """
    y = np.atleast_2d(y)
    if workers > 1 and len(y) > 1:
        from concurrent.futures import ThreadPoolExecutor
        step = max(16, -(-len(y) // (4 * workers)))
        blocks = [y[i:i + step] for i in range(0, len(y), step)]
        with ThreadPoolExecutor(workers) as pool:
            return np.concatenate(list(pool.map(lambda blk: compute_mfccs_batch(blk, precision, 1, **kw), blocks)), 0)
    m = mel_power(y, precision, **kw)
    out = np.array(m, copy=True)
    pos = out > 0
    out[pos] = np.log(out[pos])
    return dct_len1(out).astype(np.float32)


def compute_mfccs(y, precision="f64", **kw):
    """One clip, reference signature: np[n] -> np.float32 (T, n_mels, 1) (audio_processor.py:18-30)."""
    return compute_mfccs_batch(np.asarray(y)[None, :], precision, **kw)[0][:, :, None]
