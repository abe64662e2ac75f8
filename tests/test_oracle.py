"""CPU: the oracle against the reference-generated golden vectors and known answers (no GPU needed)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_model_files, load_golden_model
from oracle import frontend, models, weights


def _digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


FAST = [f for f in golden_model_files() if "hey_snips" not in f]


@pytest.mark.parametrize("fname", golden_model_files())
def test_synthetic_weights_are_reproducible(fname):
    _, _, _, sd, _, z = load_golden_model(fname)
    assert _digest(sd) == str(z["weights_sha256"])


@pytest.mark.parametrize("fname", FAST)
def test_numpy_restatement_matches_reference_logits(fname):
    # golden logits come from the reference's own model.ResNet / model.CNN (oracle/gen_golden.py)
    _, name, cfg, sd, feats, z = load_golden_model(fname)
    taps = {}
    got = models.forward_numpy(name, cfg, sd, feats, np.float64, taps)
    assert np.abs(got - z["logits"]).max() < 2e-5
    for key, val in taps.items():
        if f"tap_{key}" in z.files:
            step = int(z[f"tap_{key}_step"])
            assert np.abs(val[1].astype(np.float32).reshape(-1)[::step] - z[f"tap_{key}"]).max() < 5e-5
    assert sum(v.size for k, v in sd.items() if not k.endswith("num_batches_tracked")
               and "running_" not in k) == int(z["num_params"])


@pytest.mark.parametrize("fname", golden_model_files())
def test_torch_restatement_matches_reference_logits(fname):
    _, name, cfg, sd, feats, z = load_golden_model(fname)
    got = models.forward_torch(name, cfg, sd, feats).numpy()
    assert np.abs(got - z["logits"]).max() < 1e-5
    assert (got.argmax(1) == z["logits"].argmax(1)).all()


def test_evaluate_tail_matches_reference_evaluate():
    with open(os.path.join(GOLDEN, "evaluate_res8.json")) as f:
        g = json.load(f)
    cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    sd = weights.make_state_dict("ResNet", cfg, seed=g["weights_seed"])
    n, b = g["n_batches"], g["batch"]
    feats = weights.make_features(n * b, seed=g["seed_features"])
    labels = weights.make_labels(n * b, 12, seed=g["seed_labels"])
    logits = models.forward_torch("ResNet", cfg, sd, feats).numpy()
    loss = np.mean([models.ce_loss_numpy(logits[i * b:(i + 1) * b], labels[i * b:(i + 1) * b]) for i in range(n)])
    correct, total, pc, pt = models.accuracy_counts(logits, labels, 12)
    assert abs(loss - g["result"]["loss"]) < 1e-5
    assert abs(correct / total - g["result"]["metric_Acc"]) < 1e-12
    names = ["yes", "no", "up", "down", "left", "right", "on", "off", "stop", "go", "__unknown__", "__silence__"]
    for k, cname in enumerate(names):
        if pt[k]:
            assert abs(pc[k] / pt[k] - g["result"]["metric_PerClassAcc"][cname]) < 1e-12


# ------------------------------------------------------------------ front end
def test_dct_step_is_times_two_pinned_against_scipy():
    z = np.load(os.path.join(GOLDEN, "frontend_dct_pin.npz"))   # produced with scipy.fftpack.dct as the reference calls it
    assert np.array_equal(z["out"][:, :, 0], frontend.dct_len1(z["logmel"].T).astype(np.float32))
    import scipy.fftpack
    col = z["logmel"][:, 5:6]
    assert np.array_equal(scipy.fftpack.dct(col), 2.0 * col)


def test_frontend_known_answers():
    bank = frontend.mel_filterbank()
    assert bank.shape == (40, 241)
    nz = np.nonzero(bank.sum(axis=0))[0]
    assert nz.min() == 1 and nz.max() == 119
    assert int((bank != 0).sum()) == 230
    per = (bank != 0).sum(axis=1)
    assert per.min() == 3 and per.max() == 13
    assert frontend.num_frames(16000) == 101
    assert frontend.frame_signal(np.zeros((1, 16000))).shape == (1, 101, 480)
    assert np.array_equal(frontend.compute_mfccs_batch(np.zeros((1, 16000))), np.zeros((1, 101, 40), np.float32))
    t = np.arange(16000) / 16000.0
    tone = 0.5 * np.sin(2 * np.pi * 1000.0 * t)
    m = frontend.mel_power(tone[None])[0, 50]
    edges = frontend._mel_to_hz_slaney(np.linspace(frontend._hz_to_mel_slaney(20.0), frontend._hz_to_mel_slaney(4000.0), 42))
    peak = int(np.argmax(m))
    assert edges[peak] <= 1000.0 <= edges[peak + 2]
    out = frontend.compute_mfccs(tone)
    assert out.shape == (101, 40, 1) and out.dtype == np.float32
    # SURVEY.md Appendix A's probe values (the survey's own evaluation of the same specification, NOT librosa output: the librosa part stays unpinned):
    # band edges (item 6) and the output statistics of one second of uniform(-1, 1) noise (item 10: range ~ [-7.2, 4.9], mean ~ 0.66)
    assert np.allclose(edges[:3], [20.0, 76.689, 133.378], atol=1e-3) and np.allclose(edges[-3:], [3558.61, 3772.86, 4000.0], atol=1e-2)
    noise = np.random.default_rng(0).uniform(-1.0, 1.0, 16000).astype(np.float32)
    f = frontend.compute_mfccs_batch(noise[None], "f64")[0]
    assert abs(f.min() + 7.2) < 0.1 and abs(f.max() - 4.9) < 0.05 and abs(f.mean() - 0.66) < 0.01, (f.min(), f.max(), f.mean())


def test_frontend_reflect_padding_matches_definition():
    y = np.arange(1000, dtype=np.float64)
    fr = frontend.frame_signal(y[None], 480, 160)[0]
    assert fr[0, 0] == y[240] and fr[0, 239] == y[1] and fr[0, 240] == y[0]
    s_last = (fr.shape[0] - 1) * 160 + 479 - 240          # unpadded index of the last sample of the last frame
    assert s_last >= 1000 and fr[-1, -1] == y[2 * 999 - s_last]   # np.pad 'reflect': y[2(n-1) - s]


def test_frontend_vectors_and_f32_variant():
    z = np.load(os.path.join(GOLDEN, "frontend_vectors.npz"))
    wav = weights.make_waveforms(int(z["n_clips"]), seed=int(z["wav_seed"]))
    wav[2:6] = z["wav_special"]
    f64 = frontend.compute_mfccs_batch(wav, "f64")
    assert np.array_equal(f64, z["feats"])
    assert np.abs(frontend.mel_filterbank() - z["mel_bank"]).max() == 0
    # independent check of the STFT: torch.stft (centre, reflect, periodic Hann) on the same clips
    import torch
    spec = torch.stft(torch.from_numpy(wav).double(), 480, 160, window=torch.hann_window(480, periodic=True, dtype=torch.float64),
                      center=True, pad_mode="reflect", return_complex=True)
    p_torch = (spec.real ** 2 + spec.imag ** 2).permute(0, 2, 1).numpy()
    p_mine = frontend.power_spectrogram(wav)
    assert np.abs(p_torch - p_mine).max() <= 1e-9 * max(1.0, p_mine.max())
    # the complex64-style variant agrees with float64 wherever the band is above the fp32 noise floor
    f32 = frontend.compute_mfccs_batch(wav, "f32")
    mel = frontend.mel_power(wav, "f64")
    ok = mel > 1e-6 * mel.max(axis=(1, 2), keepdims=True)
    assert np.abs(f32 - f64)[ok].max() < 2e-3
