"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle / reference goldens."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_model_files, load_golden_model

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3        # north star: logits within 1e-3 fp32, argmax exact


@pytest.fixture(scope="module")
def torch_cuda(lib_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    torch.cuda.set_device(0)
    return torch


def _build(torch, name, cfg, sd):
    from honk2_amd.utils import find_cls
    model = find_cls(f"model.{name}")(dict(cfg))
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    return model.to("cuda:0").eval()


# ------------------------------------------------------------------ front end
def test_frontend_matches_oracle(torch_cuda):
    torch = torch_cuda
    from honk2_amd.utils import AudioProcessor
    from oracle import frontend, weights
    z = np.load(os.path.join(GOLDEN, "frontend_vectors.npz"))
    wav = weights.make_waveforms(int(z["n_clips"]), seed=int(z["wav_seed"]))
    wav[2:6] = z["wav_special"]
    got = AudioProcessor().compute_mfccs_batch(torch.from_numpy(wav).cuda()).cpu().numpy()
    want = z["feats"]                                   # float64 restatement (golden fixture)
    assert got.shape == want.shape == (14, 101, 40)
    mel = frontend.mel_power(wav, "f64")
    top = np.maximum(mel.max(axis=(1, 2), keepdims=True), 1e-30)
    err = np.abs(got - want)
    # fp32 tolerance tiers by conditioning: a band x dB below the clip's strongest band carries the rounding noise
    # of that strongest band (true of the reference's complex64 FFT as well)
    assert err[mel > 1e-4 * top].max() < 1e-3, err[mel > 1e-4 * top].max()      # within 40 dB of the peak
    assert err[mel > 1e-6 * top].max() < 1e-2, err[mel > 1e-6 * top].max()      # within 60 dB
    f32 = frontend.compute_mfccs_batch(wav, "f32")                               # complex64-style restatement
    for b in range(14):
        if b != 2:                                                               # every clip but the bare tone, see below
            assert err[b].max() < max(1e-3, 10 * np.abs(f32[b] - want[b]).max()), (b, err[b].max())
    # exact-zero clips (the silence class) must give exact zeros, not -inf
    assert np.array_equal(got[0], np.zeros((101, 40), np.float32))
    # white-noise clips and the unit impulse (clip 3: flat spectrum in the frames that hold it, exact zeros elsewhere)
    # are well conditioned everywhere: the strict bound, no tier
    for b in (3, 4, 6, 7, 8, 9, 10, 11):
        assert err[b].max() < 1e-3, (b, err[b].max())
    assert np.array_equal(got[3] == 0, want[3] == 0)                             # frames the impulse does not reach
    # Bare, bin-centred 1 kHz sine (clip 2): under the Hann window its energy sits in DFT bins 29-31; every mel band that
    # holds none of those bins is rounding noise of the peak in ANY fp32 evaluation (the complex64 restatement itself is
    # off by tens of units there), so what can be asserted is: the bands that hold the tone to the strict bound, the peak
    # where it analytically has to be, and everywhere else a rounding floor no more than 20 dB above the restatement's
    # own.
    tone_bands = np.flatnonzero((frontend.mel_filterbank()[:, 29:32] > 0).any(axis=1))
    assert list(tone_bands) == [15, 16, 17]
    assert err[2][:, tone_bands].max() < 1e-3, err[2][:, tone_bands].max()
    assert int(np.argmax(got[2, 50])) == int(np.argmax(want[2, 50])) and int(np.argmax(want[2, 50])) in tone_bands
    dev32 = np.abs(f32[2] - want[2])
    for f in range(40):     # 2 ln(power) units: the kernel's rounding floor lies at most 20 dB (2 ln 100 = 9.2) above complex64's
        assert err[2][:, f].max() <= dev32[:, f].max() + 9.3, (f, err[2][:, f].max(), dev32[:, f].max())
    assert np.isfinite(got[2]).all()


def test_frontend_reference_signature_and_odd_lengths(torch_cuda):
    torch = torch_cuda
    from honk2_amd.utils import AudioProcessor
    from oracle import frontend, weights
    ap = AudioProcessor()
    clip = weights.make_waveforms(3, seed=5)[2]
    one = ap.compute_mfccs(clip)                        # reference signature: np[n] -> (T, 40, 1) float32
    assert one.shape == (101, 40, 1) and one.dtype == np.float32
    assert np.abs(one - frontend.compute_mfccs(clip)).max() < 1e-3
    for n in (400, 16000 - 37, 16000 + 160 * 30 + 5, 16000 * 3):   # short, ragged, >112 frames (two chunks), long
        wav = weights.make_waveforms(3, n_samples=n, seed=n)[1:]
        got = ap.compute_mfccs_batch(torch.from_numpy(wav).cuda()).cpu().numpy()
        want = frontend.compute_mfccs_batch(wav, "f64")
        assert got.shape == want.shape == (2, 1 + n // 160, 40)
        err = np.abs(got - want)
        assert err[1].max() < 1e-3, n                       # white-noise clip: well conditioned everywhere
        mel = frontend.mel_power(wav, "f64")
        assert err[mel > 1e-4 * mel.max(axis=(1, 2), keepdims=True)].max() < 1e-3, n
        assert err.max() < 1e-2, n                          # dithered tone: bands ~57 dB down sit at the fp32 floor


def test_frontend_chunk_and_edge_geometry(torch_cuda):
    """16-byte-aligned lengths around the fp16 kernel's edge cases: barely longer than the reflect padding, one frame
    into a second 112-frame chunk, exactly 112 frames, and PCM input for the same lengths."""
    torch = torch_cuda
    from honk2_amd.utils import AudioProcessor
    from oracle import frontend, weights
    ap = AudioProcessor()
    for n in (244, 320, 1600, 160 * 111, 160 * 111 + 156, 160 * 112, 160 * 112 + 4, 160 * 224 + 8):
        wav = weights.make_waveforms(4, n_samples=n, seed=n)[2:]          # two white-noise clips
        want = frontend.compute_mfccs_batch(wav, "f64")
        got = ap.compute_mfccs_batch(torch.from_numpy(wav).cuda()).cpu().numpy()
        assert got.shape == want.shape == (2, 1 + n // 160, 40), n
        assert np.abs(got - want).max() < 1e-3, (n, np.abs(got - want).max())
        pcm = np.round(wav * 32767.0).astype(np.int16)
        want16 = frontend.compute_mfccs_batch(pcm.astype(np.float32) / np.float32(32768.0), "f64")
        got16 = ap.compute_mfccs_batch(torch.from_numpy(pcm).cuda()).cpu().numpy()
        assert np.abs(got16 - want16).max() < 1e-3, (n, np.abs(got16 - want16).max())


def test_frontend_kernels_agree_and_scale_with_loud_input(torch_cuda, monkeypatch):
    """Two independent front-end kernels: the default computes the DFT with three-term fp16 products (samples scaled by
    a per-chunk power of two), KWS_FRONTEND_IMPL=fp32 with the fp32-input MFMA.  Both must sit equally close to the
    float64 restatement on well-conditioned clips; scaling the samples by 2^k (up to the int16 range, far past what
    fp16 holds unscaled) must move every feature 2 ln(mel power) by 2 ln(4^k) = 4 k ln 2."""
    torch = torch_cuda
    from honk2_amd.utils import AudioProcessor
    from oracle import frontend, weights
    wav = weights.make_waveforms(12, seed=77)
    noise_like = [b for b in range(12) if b % 12 not in (0, 1)]           # skip the all-zero clip and the bare tone
    want = frontend.compute_mfccs_batch(wav, "f64")
    x = torch.from_numpy(wav).cuda()
    f16 = AudioProcessor().compute_mfccs_batch(x).cpu().numpy()
    monkeypatch.setenv("KWS_FRONTEND_IMPL", "fp32")
    f32 = AudioProcessor().compute_mfccs_batch(x).cpu().numpy()
    monkeypatch.delenv("KWS_FRONTEND_IMPL")
    e16 = np.abs(f16 - want)[noise_like].max()
    e32 = np.abs(f32 - want)[noise_like].max()
    assert e16 < 1e-3 and e32 < 1e-3, (e16, e32)
    assert e16 <= 2.0 * e32 + 1e-6, (e16, e32)                            # not a reduced-precision mode
    assert np.abs(f16 - f32)[noise_like].max() < 1e-3
    assert np.array_equal(f16[0], np.zeros_like(f16[0]))                 # silence stays exactly zero
    ap = AudioProcessor()
    base = ap.compute_mfccs_batch(x[noise_like]).cpu().numpy()
    for k in (3, 15):                                                     # |x| up to 8 and up to 32768
        loud = ap.compute_mfccs_batch(x[noise_like] * float(2 ** k)).cpu().numpy()
        assert np.isfinite(loud).all()
        assert np.abs(loud - (base + 4.0 * k * np.log(2.0))).max() < 2e-4, k


# ------------------------------------------------------------------ models against the reference goldens
@pytest.mark.parametrize("fname", golden_model_files())
def test_model_logits_match_reference(torch_cuda, fname):
    torch = torch_cuda
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    model = _build(torch, name, cfg, sd)
    got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
    want = z["logits"]
    assert got.shape == want.shape
    assert np.abs(got - want).max() < LOGIT_TOL, (tag, np.abs(got - want).max())
    assert (got.argmax(1) == want.argmax(1)).all()
    assert model.num_params() == int(z["num_params"])
    # CNNs: conv_0 from an LDS image of the clip (conv_in1.hip), conv_1 from LDS-staged bands (conv_band.hip)
    assert model.plan_name() == ("res8_fused" if tag == "resnet__res8" else "resnet_tiled" if name == "ResNet" else
                                 "cnn_band" if "conv_1" in cfg else "cnn_in1")


def test_experiment_switches_do_nothing_in_the_product_library(torch_cuda, monkeypatch):
    """A drop-in library must not compute garbage because an environment variable is set: with every ablation / timing switch of the
    EXPERIMENTS build in the environment -- KWS_R8_DEBUG=3 (skip conv_0 and the k-loops), KWS_T3_DEBUG=7 (skip k-loops, staging, stores),
    a phase-stamp file (host sync + copy inside the call) -- the product library still returns the golden logits on the fused res8, the
    tiled res15 (bf16 too: pair / triple kernels) and the cnn band plan."""
    torch = torch_cuda
    monkeypatch.setenv("KWS_R8_DEBUG", "3")
    monkeypatch.setenv("KWS_T3_DEBUG", "7")
    monkeypatch.setenv("KWS_T3_TIMING", "/tmp/kws_should_never_be_written.bin")
    monkeypatch.setenv("KWS_BAND_TIMING", "/tmp/kws_should_never_be_written.bin")
    monkeypatch.setenv("KWS_R8_WGS_PER_CU", "1")
    monkeypatch.setenv("KWS_TEST_THROW", "bad_alloc")
    for fname in ("model_resnet__res8.npz", "model_resnet__res15.npz", "model_cnn__cnn-trad-pool2.npz"):
        tag, name, cfg, sd, feats, z = load_golden_model(fname)
        got = _build(torch, name, cfg, sd)(torch.from_numpy(feats).cuda()).cpu().numpy()
        assert np.abs(got - z["logits"]).max() < LOGIT_TOL and (got.argmax(1) == z["logits"].argmax(1)).all(), fname
    tag, name, cfg, sd, feats, z = load_golden_model("model_resnet__res15.npz")
    got = _build(torch, name, dict(cfg, dtype="bf16"), sd)(torch.from_numpy(feats).cuda()).cpu().numpy()
    assert np.abs(got - z["logits"]).max() < 2e-2
    assert not os.path.exists("/tmp/kws_should_never_be_written.bin")


@pytest.mark.parametrize("fname", ["model_resnet__res15.npz", "model_cnn__cnn-trad-pool2.npz", "model_resnet__res8.npz"])
def test_reduced_precision_mode_bf16x3(torch_cuda, fname):
    """Opt-in `dtype: "bf16x3"` (BASELINE configs[2]/[4] are reduced-precision cases): three-term bf16 products,
    fp32 accumulation.  Tolerance as SURVEY.md Appendix C prescribes for fp16-class arithmetic (5e-3), argmax compared
    only where the reference's top-1/top-2 margin exceeds twice the tolerance."""
    torch = torch_cuda
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    model = _build(torch, name, dict(cfg, dtype="bf16x3"), sd)
    got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
    want = z["logits"]
    tol = 5e-3
    err = np.abs(got - want).max()
    assert err < tol, (tag, err)
    top = np.sort(want, axis=1)
    clear = (top[:, -1] - top[:, -2]) > 2 * tol
    assert (got.argmax(1) == want.argmax(1))[clear].all() and clear.any()
    exact = _build(torch, name, cfg, sd)(torch.from_numpy(feats).cuda()).cpu().numpy()
    assert np.abs(exact - want).max() <= err + 1e-6          # the default mode is at least as close to the reference


@pytest.mark.parametrize("fname", ["model_resnet__res15.npz", "model_cnn__cnn-trad-pool2.npz", "model_resnet__res8.npz",
                                   "model_resnet__res26_narrow.npz", "model_hey_snips__res26.npz", "model_resnet__res8_narrow.npz"])
def test_plain_bf16_mode(torch_cuda, fname):
    """`dtype: "bf16"` (BASELINE configs[2]: res15 bf16): operands rounded to bf16 at the matrix cores, fp32 accumulation.
    Tolerance 2e-2 at |logit| ~ 1 as SURVEY.md Appendix C prescribes; argmax compared only where the reference's
    top-1/top-2 margin exceeds twice the tolerance, and the fraction of such clips is checked to be non-trivial where
    the model separates classes at all."""
    torch = torch_cuda
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    model = _build(torch, name, dict(cfg, dtype="bf16"), sd)
    got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
    want = z["logits"]
    tol = 2e-2 * max(1.0, float(np.abs(want).max()))
    err = np.abs(got - want).max()
    assert 1e-5 < err < tol, (tag, err)          # visibly reduced precision, but inside the bf16 bar
    top = np.sort(want, axis=1)
    clear = (top[:, -1] - top[:, -2]) > 2 * tol
    assert (got.argmax(1) == want.argmax(1))[clear].all()
    x3 = _build(torch, name, dict(cfg, dtype="bf16x3"), sd)(torch.from_numpy(feats).cuda()).cpu().numpy()
    assert np.abs(x3 - want).max() < err         # the three-term mode sits between bf16 and the default


@pytest.mark.parametrize("fname", ["model_cnn__cnn-trad-pool2.npz", "model_resnet__res15.npz", "model_resnet__res8.npz",
                                   "model_hey_snips__res26.npz", "model_resnet__res15_narrow.npz", "model_cnn__cnn-one-fstride4.npz"])
def test_plain_fp16_mode(torch_cuda, fname):
    """`dtype: "fp16"` (BASELINE configs[4]: cnn-trad-pool2 fp16): operands rounded to fp16 at the matrix cores (weights carry a
    power-of-two scale), fp32 accumulation.  Tolerance 5e-3 at |logit| ~ 1 as SURVEY.md Appendix C prescribes for fp16,
    argmax compared where the reference's margin exceeds twice the tolerance; it must sit between bf16 and the default."""
    torch = torch_cuda
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    x = torch.from_numpy(feats).cuda()
    got = _build(torch, name, dict(cfg, dtype="fp16"), sd)(x).cpu().numpy()
    want = z["logits"]
    tol = 5e-3 * max(1.0, float(np.abs(want).max()))
    err = np.abs(got - want).max()
    assert 1e-6 < err < tol, (tag, err)
    top = np.sort(want, axis=1)
    clear = (top[:, -1] - top[:, -2]) > 2 * tol
    assert (got.argmax(1) == want.argmax(1))[clear].all()
    bf = _build(torch, name, dict(cfg, dtype="bf16"), sd)(x).cpu().numpy()
    assert err < np.abs(bf - want).max()


def test_res8_fused_equals_layerwise_and_oracle_on_a_larger_batch(torch_cuda, monkeypatch):
    torch = torch_cuda
    from oracle import models, weights
    cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    sd = weights.make_state_dict("ResNet", cfg, seed=11)
    feats = weights.make_features(1100, seed=12)         # > 2 clips per resident workgroup: exercises the persistent loop
    x = torch.from_numpy(feats).cuda()
    fused = _build(torch, "ResNet", cfg, sd)
    a = fused(x).cpu().numpy()
    assert fused.plan_name() == "res8_fused"
    monkeypatch.setenv("KWS_FORCE_LAYERWISE", "1")
    lw = _build(torch, "ResNet", cfg, sd)
    b = lw(x).cpu().numpy()
    assert lw.plan_name() == "resnet_tiled"
    want = models.forward_torch("ResNet", cfg, sd, feats).numpy()
    assert np.abs(a - want).max() < LOGIT_TOL and np.abs(b - want).max() < LOGIT_TOL
    assert np.abs(a - b).max() < 1e-4
    margin = np.sort(want, axis=1)
    clear = (margin[:, -1] - margin[:, -2]) > 1e-4
    assert (a.argmax(1) == want.argmax(1))[clear].all() and clear.mean() > 0.9


@pytest.mark.parametrize("fname,env,plan", [
    ("model_cnn__cnn-tpool3.npz", {"KWS_CNN_IN1": "0"}, "cnn_band"),        # 3 x 3 MaxPool in the generic kernel's channels-last epilogue
    ("model_cnn__cnn-trad-pool2.npz", {"KWS_CNN_IN1": "0"}, "cnn_band"),
    ("model_cnn__cnn-tstride8.npz", {"KWS_CNN_IN1": "0"}, "cnn_band"),      # 126 -> 128 input channels per band cell
    ("model_cnn__cnn-one-fstride8.npz", {"KWS_CNN_IN1": "0"}, "layerwise"),
    ("model_cnn__cnn-tpool2.npz", {"KWS_CNN_BAND": "0"}, "layerwise"),
])
@pytest.mark.parametrize("dtype", ["f32", "fp16"])
def test_cnn_plans_agree(torch_cuda, monkeypatch, fname, env, plan, dtype):
    """The cnn-* models run conv_0 from an LDS image of the clip (conv_in1.hip), conv_1 from LDS-staged row bands (conv_band.hip)
    and the first Linear on column-permuted weights.  KWS_CNN_IN1=0 keeps conv_0 in the generic kernel (channels-last epilogue),
    KWS_CNN_BAND=0 keeps the whole model there: every combination must meet the same bar against the reference's logits."""
    torch = torch_cuda
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    x = torch.from_numpy(feats).cuda()
    want = z["logits"]
    tol = LOGIT_TOL if dtype == "f32" else 5e-3 * max(1.0, float(np.abs(want).max()))
    default = _build(torch, name, dict(cfg, dtype=dtype), sd)
    got_default = default(x).cpu().numpy()
    assert default.plan_name() == ("cnn_band" if "conv_1" in cfg else "cnn_in1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    other = _build(torch, name, dict(cfg, dtype=dtype), sd)
    got = other(x).cpu().numpy()
    assert other.plan_name() == plan
    assert np.abs(got - want).max() < tol and np.abs(got_default - want).max() < tol, (tag, dtype)
    if dtype == "f32":
        assert np.abs(got - got_default).max() < 2e-5
        assert (got.argmax(1) == want.argmax(1)).all() and (got_default.argmax(1) == want.argmax(1)).all()


_ODD_CNNS = {
    # conv_0 with 16 kernel rows, stride (1, 2), 2 x 1 pool; 40 -> 48 channels through a 4 x 3 conv_1; straight to the labels
    "in1_band": ({"time": 81, "frequency": 48, "dropout_prob": 0.5, "n_labels": 7,
                  "conv_0": {"out_channels": 40, "kernel_size": [16, 8], "stride": [1, 2]}, "pool_0": {"kernel_size": [2, 1]},
                  "conv_1": {"out_channels": 48, "kernel_size": [4, 3], "stride": [1, 1]}, "pool_1": {"kernel_size": [1, 1]}}, "cnn_band"),
    # twelve kernel rows (three k-steps: not a shape conv_in1.hip is built for): conv_0 stays in the generic kernel and writes
    # channels-last cells from its epilogue, conv_1 on bands; 20 channels in 32-channel cells
    "generic_band": ({"time": 64, "frequency": 40, "dropout_prob": 0.5, "n_labels": 5,
                      "conv_0": {"out_channels": 20, "kernel_size": [12, 8], "stride": [1, 1]}, "pool_0": {"kernel_size": [2, 2]},
                      "conv_1": {"out_channels": 40, "kernel_size": [5, 5], "stride": [1, 1]}, "pool_1": {"kernel_size": [1, 1]},
                      "lin_0": {"out_features": 24}}, "cnn_band"),
    # conv_cols.hip off its shipped shape: 64 -> 50 channels (four tiles, the last one padded), eight kernel rows (16 sweeps), a 36 x 11 map -> 29 x 8 outputs:
    # two bands of 16 rows, the second moved up to end on the last row (three of its rows belong to the first)
    "cols_two_bands": ({"time": 88, "frequency": 40, "dropout_prob": 0.5, "n_labels": 6,
                        "conv_0": {"out_channels": 64, "kernel_size": [16, 8], "stride": [1, 1]}, "pool_0": {"kernel_size": [2, 3]},
                        "conv_1": {"out_channels": 50, "kernel_size": [8, 4], "stride": [1, 1]}, "pool_1": {"kernel_size": [1, 1]},
                        "lin_0": {"out_features": 24}}, "cnn_band"),
    # ... and at its widest: 14 output columns, exactly one band of 16 rows, twelve kernel rows
    "cols_wide": ({"time": 68, "frequency": 60, "dropout_prob": 0.5, "n_labels": 9,
                   "conv_0": {"out_channels": 60, "kernel_size": [16, 8], "stride": [2, 1]}, "pool_0": {"kernel_size": [1, 3]},
                   "conv_1": {"out_channels": 64, "kernel_size": [12, 4], "stride": [1, 1]}, "pool_1": {"kernel_size": [1, 1]}}, "cnn_band"),
    # strided conv_1: no band plan, nothing channels-last
    "strided_conv1": ({"time": 101, "frequency": 40, "dropout_prob": 0.5, "n_labels": 12,
                       "conv_0": {"out_channels": 32, "kernel_size": [20, 8], "stride": [1, 1]}, "pool_0": {"kernel_size": [2, 2]},
                       "conv_1": {"out_channels": 32, "kernel_size": [6, 4], "stride": [2, 1]}, "pool_1": {"kernel_size": [1, 1]}}, "layerwise"),
    # a 42-wide map (not a multiple of four) keeps the single-conv model off the LDS-image kernel
    "odd_width": ({"time": 50, "frequency": 42, "dropout_prob": 0.5, "n_labels": 4,
                   "conv_0": {"out_channels": 30, "kernel_size": [24, 8], "stride": [1, 2]}, "pool_0": {"kernel_size": [1, 3]},
                   "lin_0": {"out_features": 16}, "dnn_0": {"out_features": 32}}, "layerwise"),
    # 21 kernel rows padded to 24 (zero weights under uninitialised-looking rows), 3 x 3 pool, 130 channels in nine tiles
    "in1_wide": ({"time": 70, "frequency": 40, "dropout_prob": 0.5, "n_labels": 3,
                  "conv_0": {"out_channels": 130, "kernel_size": [21, 8], "stride": [2, 1]}, "pool_0": {"kernel_size": [3, 3]},
                  "lin_0": {"out_features": 32}}, "cnn_in1"),
}


@pytest.mark.parametrize("case", sorted(_ODD_CNNS))
@pytest.mark.parametrize("dtype", ["f32", "fp16"])
def test_cnn_geometries_beyond_the_shipped_configs(torch_cuda, case, dtype):
    """The cnn-* kernels (conv_in1.hip, conv_band.hip, the channels-last epilogue and the column-permuted Linear) on geometries
    none of the shipped configs has -- other map sizes, strides, pooling windows, channel counts that need padding -- each against
    the fp32 CPU oracle (reference model/cnn.py:79-107), and the plan each of them must take."""
    torch = torch_cuda
    from oracle import models, weights
    cfg, plan = _ODD_CNNS[case]
    sd = weights.make_state_dict("CNN", cfg, seed=5)
    feats = weights.make_features(37, seed=6, time=cfg["time"], freq=cfg["frequency"])
    model = _build(torch, "CNN", dict(cfg, dtype=dtype), sd)
    got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
    assert model.plan_name() == plan
    if case.startswith("cols"):      # (`fp16` tensors: the column-tile kernel; f32: three-term products stay on conv_band.hip)
        assert ("conv_cols" in model.plan_detail()) == (dtype == "fp16"), model.plan_detail()
    want = models.forward_torch("CNN", cfg, sd, feats).numpy()
    tol = LOGIT_TOL if dtype == "f32" else 5e-3 * max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() < tol, (case, dtype, np.abs(got - want).max())


def _random_resnet(rng):
    cfg = {"n_feature_maps": int(rng.choice([8, 16, 19, 24, 30, 45, 48])), "n_layers": int(rng.integers(2, 10)),
           "use_dilation": bool(rng.integers(0, 2)), "n_labels": int(rng.integers(2, 13))}
    if rng.integers(0, 3) == 0:
        cfg["pool"] = [int(rng.integers(1, 5)), int(rng.integers(1, 4))]
    return cfg, int(rng.integers(9, 131)), int(rng.integers(1, 10))


def _random_cnn(rng):
    time, freq = int(rng.integers(40, 111)), int(rng.choice([24, 32, 40, 42, 48]))
    k0 = [int(rng.integers(4, 25)), int(rng.choice([4, 8, 8, 8, 10]))]
    s0 = [int(rng.integers(1, 3)), int(rng.integers(1, 3))]
    p0 = [int(rng.integers(1, 4)), int(rng.integers(1, 4))]
    cfg = {"time": time, "frequency": freq, "dropout_prob": 0.5, "n_labels": int(rng.integers(2, 13)),
           "conv_0": {"out_channels": int(rng.choice([16, 30, 54, 64, 94])), "kernel_size": k0, "stride": s0}, "pool_0": {"kernel_size": p0}}
    h = ((time - k0[0]) // s0[0] + 1) // p0[0]
    w = ((freq - k0[1]) // s0[1] + 1) // p0[1]
    if h < 1 or w < 1:
        return None
    if rng.integers(0, 2) and h >= 6 and w >= 4:
        k1 = [int(rng.integers(2, min(h, 10) + 1)), int(rng.integers(1, min(w, 5) + 1))]
        cfg["conv_1"] = {"out_channels": int(rng.choice([24, 40, 64, 78])), "kernel_size": k1, "stride": [int(rng.integers(1, 3)), 1]}
        cfg["pool_1"] = {"kernel_size": [1, 1] if rng.integers(0, 2) else [int(rng.integers(1, 3)), int(rng.integers(1, 3))]}
        h1 = ((h - k1[0]) // cfg["conv_1"]["stride"][0] + 1) // cfg["pool_1"]["kernel_size"][0]
        w1 = ((w - k1[1]) + 1) // cfg["pool_1"]["kernel_size"][1]
        if h1 < 1 or w1 < 1:
            return None
    if rng.integers(0, 2):
        cfg["lin_0"] = {"out_features": int(rng.choice([16, 32]))}
        if rng.integers(0, 2):
            cfg["dnn_0"] = {"out_features": int(rng.choice([32, 128]))}
    return cfg


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106, 107, 108])
def test_seeded_random_model_geometries_against_the_oracle(torch_cuda, seed):
    """A seeded sweep over model configurations the reference's constructors accept (model/resnet.py:11-36, model/cnn.py:12-77) but no
    shipped config uses: channel counts, depths, dilation, pooling windows, kernel sizes, strides, clip lengths, batch sizes.  Whatever plan
    the library picks for a configuration (fused, tiled, pair, band, image, generic), the fp32 logits must match the fp32 CPU oracle."""
    torch = torch_cuda
    from oracle import models, weights
    rng = np.random.default_rng(seed)
    plans = set()
    for trial in range(8):
        cfg, T, B = _random_resnet(rng)
        sd = weights.make_state_dict("ResNet", cfg, seed=seed * 100 + trial)
        feats = (rng.standard_normal((B, T, 40)) * 2.5 + 0.65).astype(np.float32)
        model = _build(torch, "ResNet", cfg, sd)
        got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
        want = models.forward_torch("ResNet", cfg, sd, feats).numpy()
        plans.add(model.plan_name())
        assert np.abs(got - want).max() < LOGIT_TOL * max(1.0, float(np.abs(want).max())), ("ResNet", cfg, T, B, model.plan_name(), np.abs(got - want).max())
    done = 0
    while done < 8:
        cfg = _random_cnn(rng)
        if cfg is None:
            continue
        done += 1
        sd = weights.make_state_dict("CNN", cfg, seed=seed * 100 + 50 + done)
        feats = weights.make_features(int(rng.integers(1, 12)), seed=seed + done, time=cfg["time"], freq=cfg["frequency"])
        model = _build(torch, "CNN", cfg, sd)
        got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
        want = models.forward_torch("CNN", cfg, sd, feats).numpy()
        plans.add(model.plan_name())
        assert np.abs(got - want).max() < LOGIT_TOL * max(1.0, float(np.abs(want).max())), ("CNN", cfg, model.plan_name(), np.abs(got - want).max())
    assert len(plans) >= 3, plans


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_seeded_random_resnets_in_the_16_bit_dtypes(torch_cuda, dtype):
    """The same sweep for the 16-bit tensor forms of the tiled / pair kernels (and whatever falls back to the generic ones), at the
    SURVEY.md Appendix C tolerances."""
    torch = torch_cuda
    from oracle import models, weights
    rng = np.random.default_rng(211)
    for trial in range(16):
        cfg, T, B = _random_resnet(rng)
        sd = weights.make_state_dict("ResNet", cfg, seed=900 + trial)
        feats = (rng.standard_normal((B, T, 40)) * 2.5 + 0.65).astype(np.float32)
        model = _build(torch, "ResNet", dict(cfg, dtype=dtype), sd)
        got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
        want = models.forward_torch("ResNet", cfg, sd, feats).numpy()
        tol = (2e-2 if dtype == "bf16" else 5e-3) * max(1.0, float(np.abs(want).max()))
        assert np.isfinite(got).all() and np.abs(got - want).max() < tol, (cfg, T, B, dtype, model.plan_name(), np.abs(got - want).max(), np.abs(want).max())


@pytest.mark.parametrize("impl", ["nchw", "fp32"])
@pytest.mark.parametrize("fname", ["model_resnet__res15.npz", "model_resnet__res26_narrow.npz", "model_cnn__cnn-tpool2.npz"])
def test_alternative_layerwise_kernels_agree(torch_cuda, monkeypatch, fname, impl):
    """KWS_LAYERWISE_IMPL selects the generic implicit-GEMM kernels (bf16x6 over NCHW / fp32-input MFMA) instead of the
    LDS-tiled 3x3 kernel; all of them must meet the same bar."""
    torch = torch_cuda
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    monkeypatch.setenv("KWS_LAYERWISE_IMPL", impl)
    model = _build(torch, name, cfg, sd)
    got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
    assert model.plan_name() == "layerwise"
    assert np.abs(got - z["logits"]).max() < LOGIT_TOL
    assert (got.argmax(1) == z["logits"].argmax(1)).all()


@pytest.mark.parametrize("fname", ["model_resnet__res15.npz", "model_cnn__cnn-tpool2.npz"])
def test_range_free_bf16_parts_for_the_layerwise_kernels(torch_cuda, monkeypatch, fname):
    """KWS_MATRIX_PARTS=bf16: fp32-accurate products from six bf16 terms (no fp16 range limit) instead of three fp16 terms."""
    torch = torch_cuda
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    x = torch.from_numpy(feats).cuda()
    a = _build(torch, name, cfg, sd)(x).cpu().numpy()
    monkeypatch.setenv("KWS_MATRIX_PARTS", "bf16")
    b = _build(torch, name, cfg, sd)(x).cpu().numpy()
    for y in (a, b):
        assert np.abs(y - z["logits"]).max() < LOGIT_TOL and (y.argmax(1) == z["logits"].argmax(1)).all()
    assert np.abs(a - b).max() < 2e-5 and not np.array_equal(a, b)


def test_tiled_resnet_on_odd_shapes_and_unsupported_widths(torch_cuda):
    """The tiled 3x3 kernel on shapes no shipped config has: T not a multiple of anything, dilation reaching past the map,
    batches that end inside a tile, pooling; and a channel count it does not cover (falls back to the generic kernels)."""
    torch = torch_cuda
    from oracle import models, weights
    rng = np.random.default_rng(5)
    for cfg, T, B, plan in [
        ({"n_feature_maps": 45, "n_layers": 7, "use_dilation": True, "n_labels": 12}, 37, 3, "resnet_tiled"),
        ({"n_feature_maps": 19, "n_layers": 13, "use_dilation": True, "n_labels": 5}, 23, 7, "resnet_tiled"),
        ({"n_feature_maps": 42, "n_layers": 4, "use_dilation": False, "pool": [3, 2], "n_labels": 12}, 64, 2, "resnet_tiled"),
        ({"n_feature_maps": 30, "n_layers": 5, "use_dilation": True, "n_labels": 12}, 50, 4, "layerwise"),
    ]:
        sd = weights.make_state_dict("ResNet", cfg, seed=21)
        feats = (rng.standard_normal((B, T, 40)) * 2.5 + 0.65).astype(np.float32)
        model = _build(torch, "ResNet", cfg, sd)
        got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
        assert model.plan_name() == plan, (cfg, model.plan_name())
        want = models.forward_torch("ResNet", cfg, sd, feats).numpy()
        assert np.abs(got - want).max() < LOGIT_TOL, (cfg, np.abs(got - want).max())
        assert (got.argmax(1) == want.argmax(1)).all()


def test_three_fused_res8_kernels_agree_and_are_fp32_accurate(torch_cuda, monkeypatch):
    """Three independent fused kernels: the default forms fp32-accurate products from three fp16 MFMA terms (two-part fp16
    splits, weights pre-scaled by a power of two), KWS_RES8_IMPL=bf16x6 from six bf16 terms, KWS_RES8_IMPL=fp32 uses the
    fp32-input MFMA.  All must match the reference and each other, and the split-product kernels must be at least as close
    to a float64 evaluation as an fp32 implementation (torch CPU) is -- they are not reduced-precision modes."""
    torch = torch_cuda
    from oracle import models
    tag, name, cfg, sd, feats, z = load_golden_model("model_resnet__res8.npz")
    x = torch.from_numpy(feats).cuda()
    ys = {}
    for impl, plan in (("", "res8_fused"), ("bf16x6", "res8_fused_bf16x6"), ("fp32", "res8_fused_fp32mfma")):
        if impl:
            monkeypatch.setenv("KWS_RES8_IMPL", impl)
        m = _build(torch, name, cfg, sd)
        ys[plan] = m(x).cpu().numpy()
        assert m.plan_name() == plan
    for y in ys.values():
        assert np.abs(y - z["logits"]).max() < LOGIT_TOL and (y.argmax(1) == z["logits"].argmax(1)).all()
    assert np.abs(ys["res8_fused"] - ys["res8_fused_bf16x6"]).max() < 2e-5
    assert np.abs(ys["res8_fused"] - ys["res8_fused_fp32mfma"]).max() < 2e-5
    exact = models.forward_numpy(name, cfg, sd, feats, np.float64)
    err_fp32_impl = np.abs(models.forward_torch(name, cfg, sd, feats).numpy() - exact).max()
    for plan in ("res8_fused", "res8_fused_bf16x6"):
        assert np.abs(ys[plan] - exact).max() <= 2.0 * err_fp32_impl + 1e-7, (plan, np.abs(ys[plan] - exact).max(), err_fp32_impl)


def test_fused_res8_fp16_range_guard(torch_cuda):
    """A channel whose running variance is tiny multiplies activations by hundreds per layer: logits reach 1e8 and the maps
    leave fp16's range.  The fused kernel stores such maps scaled by a power of two and scales back in the consumer, so the
    result must still match fp32 arithmetic to rounding error."""
    torch = torch_cuda
    from oracle import models, weights
    cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    sd = weights.make_state_dict("ResNet", cfg, seed=5)
    for k in sd:
        if k.endswith("running_var"):
            sd[k] = np.full_like(sd[k], 2e-4)
    feats = weights.make_features(300, seed=9)
    model = _build(torch, "ResNet", cfg, sd)
    got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
    assert model.plan_name() == "res8_fused"
    want = models.forward_torch("ResNet", cfg, sd, feats).numpy()
    assert np.abs(want).max() > 1e6 and np.isfinite(got).all()
    assert np.abs(got - want).max() < 2e-6 * np.abs(want).max()


@pytest.mark.parametrize("case", ["res15", "res26_narrow", "generic30", "cnn-trad-pool2", "cnn-tpool2-fp16", "res15-fp16",
                                  "cnn-one-fpool3"])
def test_layerwise_fp16_range_guard(torch_cuda, case):
    """The layer-wise plans' default operands are two-part fp16 splits, which cannot hold |x| > 65 504, while the reference is
    fp32 throughout (model/resnet.py:48-55, model/cnn.py:79-107).  Every kernel that stores an activation notes its largest
    magnitude on the device, and a chunk that left fp16's range is recomputed on three-part bf16 operands (no environment
    knob, no host round trip).  Tiny running variances / inflated first-layer weights drive activations past 1e5 in the
    second half of an 1100-clip batch only: the result must match fp32 arithmetic to rounding error everywhere, and the
    clean first chunk must be bit-identical to what it gives on its own (its fp16 pass is kept)."""
    torch = torch_cuda
    from oracle import models, weights
    dtype = "f32"
    if case.startswith("res15"):   # "res15-fp16": fp16 TENSORS between the layers; the recomputed chunk runs on fp32 tensors
        name, cfg = "ResNet", {"n_feature_maps": 45, "n_layers": 13, "use_dilation": True, "n_labels": 12}
        dtype = "fp16" if case.endswith("fp16") else "f32"
    elif case == "res26_narrow":
        name, cfg = "ResNet", {"n_feature_maps": 19, "n_layers": 24, "use_dilation": False, "pool": [2, 2], "n_labels": 12}
    elif case == "generic30":
        name, cfg = "ResNet", {"n_feature_maps": 30, "n_layers": 6, "use_dilation": True, "n_labels": 12}
    else:
        name = "CNN"
        _, _, cfg, _, _, _ = load_golden_model({"cnn-trad-pool2": "model_cnn__cnn-trad-pool2.npz", "cnn-tpool2-fp16": "model_cnn__cnn-tpool2.npz",
                                                "cnn-one-fpool3": "model_cnn__cnn-one-fpool3.npz"}[case])
        dtype = "fp16" if case.endswith("fp16") else "f32"
    sd = weights.make_state_dict(name, cfg, seed=5)
    model = _build(torch, name, dict(cfg, dtype=dtype), sd)
    n = model.chunk_clips(100000) + 104             # a little more than the plan's largest chunk (1 024 - 4 096 clips: kws_chunk_clips)
    n_clean = model.chunk_clips(n)                  # the first chunk of this call
    assert 512 <= n_clean < n, (n_clean, n)
    n_big = 76
    feats = weights.make_features(n, seed=9)
    feats[n - n_big:] *= 40000.0                    # the last clips' inputs are large: their activations leave fp16's range, their chunk(s) are recomputed
    x = torch.from_numpy(feats).cuda()
    got = model(x).cpu().numpy()
    sel = np.r_[0:200, n_clean:n_clean + 24, n - n_big:n]      # clean clips of the kept chunk, clean clips of a recomputed one, the driven ones
    want = np.zeros_like(got)
    want[sel] = models.forward_torch(name, cfg, sd, feats[sel]).numpy()
    assert np.isfinite(got).all()
    kept, redone, driven = np.r_[0:200], np.r_[n_clean:n_clean + 24], np.r_[n - n_big:n]
    big = np.abs(want[driven]).max()
    assert big > 3e3, big                           # the driven clips really are out of the ordinary
    if dtype == "f32":
        assert np.abs(got[kept] - want[kept]).max() < LOGIT_TOL
    else:                                           # plain fp16 operands: the fp16 bar on the kept chunk, fp32 accuracy on the recomputed one
        assert np.abs(got[kept] - want[kept]).max() < 5e-3 * max(1.0, np.abs(want[kept]).max())
    assert np.abs(got[redone] - want[redone]).max() < max(LOGIT_TOL, 3e-6 * np.abs(want[redone]).max())      # (ResNet cases: these are driven clips too)
    assert np.abs(got[driven] - want[driven]).max() < 3e-6 * big, (np.abs(got[driven] - want[driven]).max(), big)
    assert torch.equal(model(x[:n_clean]), torch.from_numpy(got[:n_clean]).cuda())     # the clean chunk kept its fp16 pass


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", ["res15", "res26", "hey_snips"])
def test_layer_pairs_in_one_kernel_are_bit_identical(torch_cuda, dtype, case, monkeypatch):
    """16-bit tensors on the tiled plan: an odd layer and the even layer behind it run in ONE kernel when they share a dilation
    (conv_i's output stays in LDS, x_{i-1} serves as input and residual from the same staged tile; reference model/resnet.py:46-55).
    The intermediate map is rounded exactly as the store it replaces, so the logits must equal the one-kernel-per-layer form bit for
    bit -- on res15 (pairs at dilations 1, 2, 4, 8), res26 (pooled map, every pair) and the 901-frame hey_snips model (dilations to 128,
    partial sub-maps), over a batch that spans several workgroups and a chunk boundary."""
    torch = torch_cuda
    from oracle import weights
    if case == "res15":
        cfg, n, t = {"n_feature_maps": 45, "n_layers": 13, "use_dilation": True, "n_labels": 12}, 1100, 101
    elif case == "res26":
        cfg, n, t = {"n_feature_maps": 45, "n_layers": 24, "use_dilation": False, "pool": [2, 2], "n_labels": 12}, 300, 101
    else:
        cfg, n, t = {"n_feature_maps": 45, "n_layers": 24, "use_dilation": True, "n_labels": 2}, 9, 901
    monkeypatch.setenv("KWS_T3_STREAM", "0")      # the tile / pair / triple kernels themselves (the streams of conv3x3_stream.hip have their own test below)
    sd = weights.make_state_dict("ResNet", cfg, seed=11)
    x = torch.from_numpy(weights.make_features(n, seed=12, time=t)).cuda()
    monkeypatch.setenv("KWS_T3_PAIR", "1")
    fused = _build(torch, "ResNet", dict(cfg, dtype=dtype), sd)
    got = fused(x)
    assert fused.plan_name() == "resnet_tiled"
    monkeypatch.setenv("KWS_T3_PAIR", "0")
    plain = _build(torch, "ResNet", dict(cfg, dtype=dtype), sd)
    want = plain(x)
    assert torch.isfinite(got).all() and torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", ["res15", "res26", "hey_snips", "short"])
def test_layer_triples_in_one_kernel_are_bit_identical(torch_cuda, dtype, case, monkeypatch):
    """16-bit tensors on the tiled plan: a run of THREE layers of one dilation runs in one kernel (conv3x3_triple_kernel; reference
    model/resnet.py:20-26, 46-55) -- first layer even (residual from memory in its own layout, x_a kept in LDS as the third layer's
    residual) or odd (residual = the staged input, x_{a+1} also stored for the next even layer).  Both intermediate maps are rounded
    as the stores they replace: logits equal the pairs-and-singles form AND the one-kernel-per-layer form bit for bit.  res15 (runs
    at dilations 2, 4, 8 by default, 1 too wide for the LDS), res26 (one run of 24 layers: pairs stay), hey_snips (dilations to 128, partial sub-maps, 901 frames), and a 37-frame, 12-layer
    model whose maps are smaller than a workgroup tile; batches that span several workgroups and a chunk boundary."""
    torch = torch_cuda
    from oracle import weights
    if case == "res15":
        cfg, n, t = {"n_feature_maps": 45, "n_layers": 13, "use_dilation": True, "n_labels": 12}, 1100, 101
    elif case == "res26":
        cfg, n, t = {"n_feature_maps": 45, "n_layers": 24, "use_dilation": False, "pool": [2, 2], "n_labels": 12}, 300, 101
    elif case == "hey_snips":
        cfg, n, t = {"n_feature_maps": 45, "n_layers": 24, "use_dilation": True, "n_labels": 2}, 9, 901
    else:
        cfg, n, t = {"n_feature_maps": 45, "n_layers": 12, "use_dilation": True, "n_labels": 12}, 3, 37
    monkeypatch.setenv("KWS_T3_STREAM", "0")      # the tile / pair / triple kernels themselves (the streams of conv3x3_stream.hip have their own test below)
    sd = weights.make_state_dict("ResNet", cfg, seed=11)
    x = torch.from_numpy(weights.make_features(n, seed=12, time=t)).cuda()
    outs, plans = {}, {}
    for tag, pair, triple in (("triples", "1", "1"), ("any_three", "1", "2"), ("pairs", "1", "0"), ("singles", "0", "0")):
        monkeypatch.setenv("KWS_T3_PAIR", pair)
        monkeypatch.setenv("KWS_T3_TRIPLE", triple)
        m = _build(torch, "ResNet", dict(cfg, dtype=dtype), sd)
        outs[tag] = m(x)
        assert m.plan_name() == "resnet_tiled"
        plans[tag] = m.plan_detail()
    # what ran: the default takes runs of exactly three (res15: dilations 2, 4, 8; dilation 1's halo rows do not fit the LDS), "any_three" every
    # three consecutive layers of one dilation
    assert "triple" not in plans["pairs"] and "pair(" in plans["pairs"] and "pair" not in plans["singles"], plans
    if case == "res15":
        assert plans["triples"] == "conv0 pair(1,2) conv(3) triple(4,5,6) triple(7,8,9) triple(10,11,12) conv(13) mean+linear", plans
    elif case == "res26":
        assert "triple" not in plans["triples"], plans      # (a run of 24; with 20-cell rows an odd-first run's three maps exceed the LDS, so "any_three" stays on pairs too)
    elif case == "hey_snips":
        assert plans["triples"].count("triple") == 7 and "pair(1,2) conv(3)" in plans["triples"], plans
    else:
        assert plans["triples"].count("triple") >= 3, plans
    assert torch.isfinite(outs["singles"]).all()
    for tag in ("triples", "any_three", "pairs"):
        assert torch.equal(outs[tag], outs["singles"]), (tag, float((outs[tag] - outs["singles"]).abs().max()))


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", ["res15", "res26", "res26_4096", "hey_snips", "short", "tiny", "pooled", "five"])
def test_layer_streams_are_bit_identical(torch_cuda, dtype, case, monkeypatch):
    """16-bit tensors, 41-48 channels (round 5, conv3x3_stream.hip): any three consecutive layers of one dilation, and any single layer, run as
    a persistent weight-stationary stream -- a layer's weights in one wave's registers, the three layers as three waves 112 positions apart on
    LDS rings, a loader wave in front (reference model/resnet.py:20-26, 44-56).  Same fragments, same K order, same epilogue, same rounding of
    the intermediate maps: the logits must equal the one-kernel-per-layer tile form bit for bit -- res15 (runs at dilations 1, 2, 4, 8 + layer
    13 alone at 16), res26 (eight runs on the pooled map), hey_snips (901 frames, dilations to 128: sub-maps three cells wide), a 37-frame
    12-layer model, one 23-frame clip (spans shorter than the pipeline), a (3, 2)-pooled map and a five-layer model (a run + two singles); batches
    that span many workgroups and a chunk boundary."""
    torch = torch_cuda
    from oracle import weights
    cfg, n, t = {
        "res15": ({"n_feature_maps": 45, "n_layers": 13, "use_dilation": True, "n_labels": 12}, 1100, 101),
        "res26": ({"n_feature_maps": 45, "n_layers": 24, "use_dilation": False, "pool": [2, 2], "n_labels": 12}, 300, 101),
        # 4 096 clips = ONE launch of 4.1 M cells, spans of exactly 250 steps: where positions of a straddling tile beyond what a layer needs once fed
        # ring garbage to the fp16 range guard (a spurious second pass: logits 1.7e-3 off the single-pass ones, 18.5 ms instead of 6)
        "res26_4096": ({"n_feature_maps": 45, "n_layers": 24, "use_dilation": False, "pool": [2, 2], "n_labels": 12}, 4096, 101),
        "hey_snips": ({"n_feature_maps": 45, "n_layers": 24, "use_dilation": True, "n_labels": 2}, 9, 901),
        "short": ({"n_feature_maps": 45, "n_layers": 12, "use_dilation": True, "n_labels": 12}, 3, 37),
        "tiny": ({"n_feature_maps": 45, "n_layers": 6, "use_dilation": True, "n_labels": 12}, 1, 23),
        "pooled": ({"n_feature_maps": 41, "n_layers": 8, "use_dilation": False, "pool": [3, 2], "n_labels": 12}, 70, 64),
        "five": ({"n_feature_maps": 48, "n_layers": 5, "use_dilation": False, "n_labels": 7}, 33, 50),
    }[case]
    sd = weights.make_state_dict("ResNet", cfg, seed=11)
    x = torch.from_numpy(weights.make_features(n, seed=12, time=t)).cuda()
    monkeypatch.setenv("KWS_T3_STREAM", "2")          # every run of three layers and every odd single layer, whatever the launch size (the default, "1", leaves launches of < 7 680 cells per CU to the tile kernels)
    fused = _build(torch, "ResNet", dict(cfg, dtype=dtype), sd)
    got = fused(x)
    assert fused.plan_name() == "resnet_tiled"
    plan = fused.plan_detail()
    if case == "res15":
        assert plan == "conv0 stream(1,2,3) stream(4,5,6) stream(7,8,9) stream(10,11,12) stream(13) mean+linear", plan
    elif case in ("res26", "res26_4096", "hey_snips"):
        assert plan.count("stream(") == 8 and "pair" not in plan and "triple" not in plan and "conv(" not in plan, plan
    elif case == "five":
        assert plan == "conv0 stream(1,2,3) conv(4) stream(5) mean+linear", plan        # (an even single layer stays on the tile kernel)
    monkeypatch.setenv("KWS_T3_STREAM", "0")
    monkeypatch.setenv("KWS_T3_PAIR", "0")
    plain = _build(torch, "ResNet", dict(cfg, dtype=dtype), sd)
    want = plain(x)
    assert "stream" not in plain.plan_detail() and "pair" not in plain.plan_detail()
    assert torch.isfinite(got).all() and torch.equal(got, want), (case, dtype, float((got - want).abs().max()))
    assert torch.equal(fused(x[: max(1, n // 3)]), want[: max(1, n // 3)])          # a clip's logits do not depend on its batch
    # odd-first runs only as streams, the tile kernels' triples / pairs for the rest (the round's intermediate plan: still selectable) -- same bits
    monkeypatch.setenv("KWS_T3_STREAM", "3")
    monkeypatch.setenv("KWS_T3_PAIR", "1")
    mixed = _build(torch, "ResNet", dict(cfg, dtype=dtype), sd)
    assert torch.equal(mixed(x), want)
    if case == "res15":
        assert mixed.plan_detail() == "conv0 stream(1,2,3) triple(4,5,6) stream(7,8,9) triple(10,11,12) stream(13) mean+linear", mixed.plan_detail()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_layer_pairs_on_odd_shapes(torch_cuda, dtype, monkeypatch):
    """The pair kernel where no shipped config puts it: maps smaller than one workgroup tile (a batch of one 23-frame clip), a map whose
    flattened positions end inside a tile, a pooled map, dilations that reach past the map, 19 and 45 channels.  Bit-identical to the
    one-kernel-per-layer form, and within the 16-bit tolerance of the fp32 oracle (reference model/resnet.py:38-60)."""
    torch = torch_cuda
    from oracle import models, weights
    monkeypatch.setenv("KWS_T3_STREAM", "0")
    rng = np.random.default_rng(31)
    for cfg, T, B in [
        ({"n_feature_maps": 45, "n_layers": 6, "use_dilation": True, "n_labels": 12}, 23, 1),
        ({"n_feature_maps": 45, "n_layers": 13, "use_dilation": True, "n_labels": 12}, 37, 3),
        ({"n_feature_maps": 19, "n_layers": 8, "use_dilation": False, "n_labels": 5}, 61, 5),
        ({"n_feature_maps": 45, "n_layers": 8, "use_dilation": False, "pool": [3, 2], "n_labels": 12}, 64, 2),
    ]:
        sd = weights.make_state_dict("ResNet", cfg, seed=23)
        feats = (rng.standard_normal((B, T, 40)) * 2.5 + 0.65).astype(np.float32)
        x = torch.from_numpy(feats).cuda()
        monkeypatch.setenv("KWS_T3_PAIR", "1")
        fused = _build(torch, "ResNet", dict(cfg, dtype=dtype), sd)
        got = fused(x)
        assert fused.plan_name() == "resnet_tiled", (cfg, fused.plan_name())
        monkeypatch.setenv("KWS_T3_PAIR", "0")
        want = _build(torch, "ResNet", dict(cfg, dtype=dtype), sd)(x)
        assert torch.isfinite(got).all() and torch.equal(got, want), (cfg, float((got - want).abs().max()))
        ref = models.forward_torch("ResNet", cfg, sd, feats).numpy()
        err = np.abs(got.cpu().numpy() - ref).max()
        assert err < (2e-2 if dtype == "bf16" else 5e-3) * max(1.0, np.abs(ref).max()), (cfg, dtype, err)


def test_fused_res8_takes_any_fp32_feature_range(torch_cuda):
    """The reference's ResNet.forward takes any finite fp32 feature (model/resnet.py:39-41); the fused res8 kernel stages features
    as fp16 pairs.  kws_forward therefore measures every clip's feature range on the device and the kernel stages out-of-range
    clips scaled down by a power of two that conv_0's pooling multiplier undoes (exact): clips whose features were blown up by
    40 000 (and by 3e7, beyond fp16's 65 504 many times over) match fp32 arithmetic to rounding error, clean clips in the same
    batch stay bit-identical to running them alone, and the fused plan is the one that ran."""
    torch = torch_cuda
    from oracle import models, weights
    cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    sd = weights.make_state_dict("ResNet", cfg, seed=5)
    n_clean, n = 600, 1100
    feats = weights.make_features(n, seed=9)
    feats[n_clean:900] *= 40000.0
    feats[900:] *= 3.0e7
    model = _build(torch, "ResNet", cfg, sd)
    x = torch.from_numpy(feats).cuda()
    got = model(x).cpu().numpy()
    assert model.plan_name() == "res8_fused"
    want = models.forward_torch("ResNet", cfg, sd, feats).numpy()
    assert np.isfinite(got).all()
    assert np.abs(got[:n_clean] - want[:n_clean]).max() < LOGIT_TOL
    for lo, hi in ((n_clean, 900), (900, n)):
        big = np.abs(want[lo:hi]).max()
        assert big > 3e3, big
        assert np.abs(got[lo:hi] - want[lo:hi]).max() < 3e-6 * big, (np.abs(got[lo:hi] - want[lo:hi]).max(), big)
        assert (got[lo:hi].argmax(1) == want[lo:hi].argmax(1)).mean() > 0.99
    assert torch.equal(model(x[:n_clean]), torch.from_numpy(got[:n_clean]).cuda())


def test_wav_to_logits_end_to_end(torch_cuda):
    torch = torch_cuda
    from oracle import frontend, models, weights
    cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    sd = weights.make_state_dict("ResNet", cfg, seed=7)
    model = _build(torch, "ResNet", cfg, sd)
    wav = weights.make_waveforms(48, seed=1234)
    got = model.forward_wav(torch.from_numpy(wav).cuda()).cpu().numpy()
    want = models.forward_numpy("ResNet", cfg, sd, frontend.compute_mfccs_batch(wav, "f64"), np.float64)
    assert np.abs(got - want).max() < LOGIT_TOL, np.abs(got - want).max()
    assert (got.argmax(1) == want.argmax(1)).all()
    # two-call form (kws_mfcc + kws_forward) gives the same logits as the fused call
    from honk2_amd.utils import AudioProcessor
    feats = AudioProcessor().compute_mfccs_batch(torch.from_numpy(wav).cuda())
    assert np.abs(model(feats).cpu().numpy() - got).max() < 1e-5


def test_pcm16_input_with_noise_mix(torch_cuda):
    """16-bit PCM input path (x/32768 [+ noise*pct], dataset/gsc_dataset.py:163-174) fused into the front end."""
    torch = torch_cuda
    from honk2_amd.utils import AudioProcessor
    from oracle import frontend, models, weights
    cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    sd = weights.make_state_dict("ResNet", cfg, seed=7)
    model = _build(torch, "ResNet", cfg, sd)
    rng = np.random.default_rng(3)
    pcm = np.clip(rng.normal(0, 3000, size=(24, 16000)), -32768, 32767).astype(np.int16)
    pcm[5] = 0                                                    # digital silence
    noise = (0.05 * rng.standard_normal((24, 16000))).astype(np.float32)
    pct = np.float32(0.1)
    as_float = pcm.astype(np.float32) / np.float32(32768.0)
    mixed = as_float + noise * pct                                # float32 arithmetic, as the reference's `data += ...`
    d_pcm, d_noise = torch.from_numpy(pcm).cuda(), torch.from_numpy(noise).cuda()
    # no noise: identical to feeding the converted floats (the conversion is exact)
    a = model.forward_wav(d_pcm)
    b = model.forward_wav(torch.from_numpy(as_float).cuda())
    assert torch.equal(a, b)
    # with noise: identical to feeding the mixed floats, and within tolerance of the oracle
    c = model.forward_wav(d_pcm, noise=d_noise, noise_pct=float(pct))
    d = model.forward_wav(torch.from_numpy(mixed).cuda())
    assert torch.equal(c, d)
    want = models.forward_numpy("ResNet", cfg, sd, frontend.compute_mfccs_batch(mixed, "f64"), np.float64)
    assert np.abs(c.cpu().numpy() - want).max() < LOGIT_TOL and (c.cpu().numpy().argmax(1) == want.argmax(1)).all()
    feats = AudioProcessor().compute_mfccs_batch(d_pcm, noise=d_noise, noise_pct=float(pct)).cpu().numpy()
    assert np.abs(feats - frontend.compute_mfccs_batch(mixed, "f64")).max() < 1e-3
    assert np.array_equal(AudioProcessor().compute_mfccs_batch(d_pcm)[5].cpu().numpy(), np.zeros((101, 40), np.float32))


def test_size_independent_properties_at_full_batch(torch_cuda):
    """BASELINE size (8192 clips/GPU): batch-composition independence and repeatability."""
    torch = torch_cuda
    from oracle import weights
    cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    model = _build(torch, "ResNet", cfg, weights.make_state_dict("ResNet", cfg, seed=7))
    g = torch.Generator(device="cuda").manual_seed(5)
    wav = (0.1 * torch.randn(8192, 16000, generator=g, device="cuda")).clamp_(-1, 1)
    wav[::12] = 0
    full = model.forward_wav(wav)
    again = model.forward_wav(wav)
    assert torch.equal(full, again)                                 # deterministic
    perm = torch.randperm(8192, device="cuda", generator=g)
    assert torch.equal(model.forward_wav(wav[perm]), full[perm])    # clips are independent units
    assert torch.equal(model.forward_wav(wav[100:357]), full[100:357])   # ragged sub-batch
    assert torch.equal(full[0], full[12]) and torch.isfinite(full).all()  # all-zero clips agree


@pytest.mark.parametrize("fname", ["model_resnet__res8.npz", "model_resnet__res15.npz", "model_cnn__cnn-trad-pool2.npz"])
def test_forward_wav_is_graph_capturable_after_one_warm_up_call(torch_cuda, fname):
    """include/kws.h: a compute call does no host synchronisation and no allocation once the first call per clip length has run, so
    a call sequence can be captured into a hipGraph.  Capture wav -> logits (front end + model, the counter resets between them)
    and mfcc alone, replay on new input, compare with the eager calls bit for bit."""
    torch = torch_cuda
    from oracle import weights
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    model = _build(torch, name, cfg, sd)
    wav = torch.from_numpy(weights.make_waveforms(48, seed=5)).cuda()
    wav2 = torch.from_numpy(weights.make_waveforms(48, seed=6)).cuda()
    static_in = wav.clone()
    out = torch.empty((48, cfg["n_labels"]), dtype=torch.float32, device="cuda")
    want1 = model.forward_wav(wav).clone()            # the warm-up call: parameters finalised, tables of this clip length built
    want2 = model.forward_wav(wav2).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        model.forward_wav(static_in, out=out)          # (on the capture stream once, un-captured)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            model.forward_wav(static_in, out=out)
    torch.cuda.current_stream().wait_stream(side)
    for src, want in ((wav2, want2), (wav, want1), (wav2, want2)):
        static_in.copy_(src)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), float((out - want).abs().max())


def test_literal_tone_clip_logits_across_four_front_ends(torch_cuda):
    """SURVEY.md 8(d)'s tone clip exactly as specified (0.5 sin(2 pi 1000 t), no dither).  A bin-centred sine leaves 90 % of the mel bands
    pure rounding noise, so the features of those bands -- and through them the logits -- differ between ANY two implementations, the
    oracle's own complex64-style and float64 variants included (bench.py records the numbers: `parity_literal_tone`).  What has to hold:
    every pair agrees on the bands that carry signal and on the argmax, and the default three-term fp16 front end is not the outlier --
    its logits are no further from the complex64-style restatement than the float64 restatement's are."""
    torch = torch_cuda
    import bench
    model, sd = bench.build_model(torch, torch.device("cuda", 0))
    rec = bench.literal_tone_record(torch, model, sd, torch.device("cuda", 0))
    pairs = rec["pairs"]
    assert len(pairs) == 6 and rec["bands_within_40dB"] < 0.2 * rec["bands_total"]
    for name, p in pairs.items():
        assert p["argmax_equal"], name
        assert p["max_abs_dfeature_within_40dB_of_peak"] < 1e-4, (name, p)
    ref = pairs["oracle_f32 vs oracle_f64"]["max_abs_dlogit"]
    assert ref > 1e-3                                                     # the clip is ill-conditioned for everybody
    assert pairs["gpu_f16x3 vs oracle_f32"]["max_abs_dlogit"] <= ref
    assert pairs["gpu_f16x3 vs oracle_f32"]["max_abs_dlogit"] <= 2.0 * pairs["gpu_fp32_mfma vs oracle_f32"]["max_abs_dlogit"] + 1e-3


def test_real_data_parallel_replicas_compute(torch_cuda):
    """torch.nn.parallel.replicate() itself (the reference's multi-GPU mechanism is nn.DataParallel, run/test.py:69-70), twice onto
    the one GPU this box has: replicas have no parameters() and a state_dict() without weights, and must still compute the
    module's logits through the engine of their device."""
    torch = torch_cuda
    tag, name, cfg, sd, feats, z = load_golden_model("model_resnet__res8.npz")
    model = _build(torch, name, cfg, sd)
    x = torch.from_numpy(feats).cuda()
    want = model(x)
    try:
        replicas = torch.nn.parallel.replicate(model, [0, 0])
    except Exception as exc:   # noqa: BLE001
        pytest.skip(f"replicate() onto one device twice is refused here: {exc!r}")
    for r in replicas:
        assert list(r.parameters()) == []
        assert torch.equal(r(x), want)
    assert len(model._engines) == 1


def test_absurd_descriptions_come_back_as_error_codes(torch_cuda):
    """include/kws.h: nothing throws across the ABI, and no size is allocated from before it is bounded: a depth of INT_MAX (whose
    `+ 1` would wrap, or resize a vector to terabytes) or a width of 2^30 comes back as a code and a message."""
    import ctypes
    from honk2_amd import _lib
    lib = _lib.load()
    for kw in (dict(n_layers=2 ** 31 - 1, n_feature_maps=45), dict(n_layers=6, n_feature_maps=2 ** 30), dict(n_layers=6, n_feature_maps=45, freq=2 ** 30)):
        h = ctypes.c_void_p()
        d = _lib.make_desc(_lib.KWS_MODEL_RESNET, n_labels=12, pool_h=4, pool_w=3, **kw)
        rc = lib.kws_create(ctypes.byref(d), ctypes.byref(h))
        assert rc == _lib.KWS_EUNSUPPORTED and not h.value and b"out of range" in lib.kws_last_error(), (kw, rc)
        with pytest.raises(RuntimeError):
            _lib.Engine(d)


def test_graph_replays_back_to_back_without_synchronisation(torch_cuda):
    """Round 3 saw a GPU memory fault on un-synchronised back-to-back replays of the captured wav -> logits graph, when each
    persistent kernel had a memset node in front of it that re-armed its work queue.  The queues now re-arm themselves (the
    last workgroup to retire zeroes them: kws_internal.h queue_retire), so a replay is two kernel nodes and nothing else.  1 500
    clips (more than the 512 resident workgroups, so the queue hands out most of them -- the 48-clip capture test never touches
    it): four replays on new inputs with NO host synchronisation in between, each compared bit for bit with the eager call."""
    torch = torch_cuda
    from oracle import weights
    tag, name, cfg, sd, feats, z = load_golden_model("model_resnet__res8.npz")
    model = _build(torch, name, cfg, sd)
    n = 1500
    srcs = [torch.from_numpy(weights.make_waveforms(n, seed=40 + i)).cuda() for i in range(2)]
    wants = [model.forward_wav(w).clone() for w in srcs]
    assert model.plan_name() == "res8_fused"
    static_in = srcs[0].clone()
    out = torch.empty((n, cfg["n_labels"]), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        model.forward_wav(static_in, out=out)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            model.forward_wav(static_in, out=out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    order = [1, 0, 1, 1]
    outs = [torch.empty_like(out) for _ in order]
    for slot, k in zip(outs, order):          # copy in, replay, copy out: stream-ordered, the host never waits
        static_in.copy_(srcs[k])
        graph.replay()
        slot.copy_(out)
    torch.cuda.synchronize()
    for slot, k in zip(outs, order):
        assert torch.equal(slot, wants[k]), float((slot - wants[k]).abs().max())
    assert torch.equal(model.forward_wav(srcs[0]), wants[0])      # and the eager path finds the queues as it left them


def test_captured_features_path_survives_a_larger_eager_batch(torch_cuda):
    """kws_forward on the fused res8 plan keeps 4 bytes per clip (the feature-range shifts) in a buffer of the handle that grows with the
    largest batch seen.  A graph captured at a SMALL batch has that buffer's address baked into two kernel nodes; an eager call with a
    larger batch then needs a bigger block.  The outgrown block must stay alive (it is parked until kws_destroy, include/kws.h), or the
    next replay writes freed memory.  Capture at 64 clips, run eagerly at 4 096, replay: bit-identical to the eager 64-clip result --
    also for a clip whose features need a shift.  And a call that would have to allocate DURING capture fails with a code, it does
    not allocate."""
    torch = torch_cuda
    from oracle import weights
    tag, name, cfg, sd, feats, z = load_golden_model("model_resnet__res8.npz")
    model = _build(torch, name, cfg, sd)
    small = torch.from_numpy(weights.make_features(64, seed=31)).cuda()
    small[5] *= 40000.0                                  # beyond fp16: the shift buffer is really used
    big = torch.from_numpy(weights.make_features(4096, seed=32)).cuda()
    want_small = model(small).clone()
    assert model.plan_name() == "res8_fused"
    static_in = small.clone()
    out = torch.empty((64, cfg["n_labels"]), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        model(static_in, out=out)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            model(static_in, out=out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    want_big = model(big).clone()                        # grows the handle's buffer: a new block, the captured one parked
    torch.cuda.synchronize()
    junk = torch.full((1 << 20,), 7, dtype=torch.int32, device="cuda")    # whatever the allocator hands out next must not be what the graph writes
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want_small), float((out - want_small).abs().max())
    assert bool((junk == 7).all()) and torch.equal(model(big), want_big)
    # allocation under capture is refused with a code (the capture itself is abandoned by the caller)
    huge = torch.from_numpy(weights.make_features(9000, seed=33)).cuda()
    out_h = torch.empty((9000, cfg["n_labels"]), dtype=torch.float32, device="cuda")
    model.forward_wav(torch.zeros((9000, 16000), device="cuda"))      # the Python host's workspace is large enough now; the library's shift buffer is not
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with pytest.raises(RuntimeError, match="warm-up"):
            with torch.cuda.graph(g2, stream=side):
                model(huge, out=out_h)
    torch.cuda.synchronize()
    assert torch.equal(model(small), want_small)         # the handle is still usable


def test_hundred_back_to_back_replays_at_the_shard_size(torch_cuda):
    """The round-3 observation at its own scale, once: 100 replays of the captured wav -> logits graph of an 8 192-clip shard, launched
    back to back with no host synchronisation.  Then each replay took 0.14 ms instead of 1.7 (most workgroups found the work queue
    already drained) and the run ended in a memory fault.  Now a replay holds two kernel nodes whose queues re-arm themselves: the
    replays must serialise (time per replay close to the eager call's, not a twelfth of it), the last one must leave the eager
    result bit for bit, and the queue words must be back at zero for the eager call that follows."""
    torch = torch_cuda
    from oracle import weights
    tag, name, cfg, sd, feats, z = load_golden_model("model_resnet__res8.npz")
    model = _build(torch, name, cfg, sd)
    n, reps = 8192, 100
    base = torch.from_numpy(weights.make_waveforms(1024, seed=77)).cuda()
    wav = base.repeat(8, 1) * torch.linspace(0.5, 1.0, n, device="cuda")[:, None]      # 8 192 distinct clips
    want = model.forward_wav(wav).clone()
    assert model.plan_name() == "res8_fused"
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(10):
        model.forward_wav(wav)
    ev[1].record()
    torch.cuda.synchronize()
    eager_ms = ev[0].elapsed_time(ev[1]) / 10
    static_in = wav.clone()
    out = torch.empty((n, cfg["n_labels"]), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        model.forward_wav(static_in, out=out)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            model.forward_wav(static_in, out=out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    out.zero_()
    ev[2].record()
    for _ in range(reps):
        graph.replay()
    ev[3].record()
    torch.cuda.synchronize()
    replay_ms = ev[2].elapsed_time(ev[3]) / reps
    assert torch.equal(out, want), float((out - want).abs().max())
    assert replay_ms > 0.6 * eager_ms, (replay_ms, eager_ms)        # every replay did all of its work, one after the other
    assert torch.equal(model.forward_wav(wav), want)
    print({"eager_ms": round(eager_ms, 3), "replay_ms": round(replay_ms, 3)})


@pytest.mark.parametrize("fname", ["model_resnet__res8.npz", "model_cnn__cnn-tpool2.npz"])
def test_plain_c_client_of_the_abi(torch_cuda, fname, tmp_path):
    """The drop-in boundary is a C ABI: tests/c_abi/kws_c_client.c (gcc, libkws_hip.so + the HIP runtime, no Python, no PyTorch; device
    memory and a stream of its own) loads the same state dict through kws_load_weights and runs kws_forward_wav / kws_forward_pcm16.
    Its logits must equal the Python host's bit for bit (the host is plumbing, not arithmetic)."""
    import ctypes
    import shutil
    import struct
    import subprocess
    torch = torch_cuda
    from oracle import weights
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / HIP headers on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "kws_c_client")
    libdir = os.path.join(root, "honk2_amd")
    build = subprocess.run(["gcc", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                            os.path.join(root, "tests", "c_abi", "kws_c_client.c"), "-o", exe, "-L" + libdir, "-lkws_hip", "-L/opt/rocm/lib",
                            "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    model = _build(torch, name, cfg, sd)
    wav = weights.make_waveforms(40, seed=9)
    bundle = tmp_path / "bundle.bin"
    with open(bundle, "wb") as f:
        f.write(bytes(model._make_desc()))
        f.write(struct.pack("<i", len(sd)))
        for key, val in sd.items():
            if key.endswith("num_batches_tracked"):
                val = np.zeros(1, np.float32)            # (accepted and ignored by kws_load_weights)
            data = np.ascontiguousarray(val, dtype=np.float32).tobytes()
            f.write(struct.pack("<i", len(key)) + key.encode() + struct.pack("<q", len(data)) + data)
        f.write(struct.pack("<ii", wav.shape[0], wav.shape[1]) + np.ascontiguousarray(wav, dtype=np.float32).tobytes())
    want = model.forward_wav(torch.from_numpy(wav).cuda()).cpu().numpy()
    pcm = np.clip(np.where(wav < 0, wav * 32768.0 - 0.5, wav * 32768.0 + 0.5), -32768, 32767).astype(np.int16)   # the client's rounding
    want_pcm = model.forward_wav(torch.from_numpy(pcm).cuda()).cpu().numpy()
    for extra, ref in (([], want), (["pcm"], want_pcm)):
        out = tmp_path / ("logits_pcm.bin" if extra else "logits.bin")
        run = subprocess.run([exe, str(bundle), str(out)] + extra, capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, (run.stdout, run.stderr[-2000:])
        assert model.plan_name() in run.stdout
        got = np.fromfile(out, dtype=np.float32).reshape(ref.shape)
        assert np.array_equal(got, ref), np.abs(got - ref).max()


def test_two_handles_on_two_streams_are_independent(torch_cuda):
    """include/kws.h: distinct handles are independent.  Two models (a fused res8 and a tiled res15, each with its own work-queue words,
    workspace and range-guard state) run wav -> logits concurrently on two streams; every result must equal the one computed alone."""
    torch = torch_cuda
    from oracle import weights
    _, name8, cfg8, sd8, _, _ = load_golden_model("model_resnet__res8.npz")
    _, name15, cfg15, sd15, _, _ = load_golden_model("model_resnet__res15.npz")
    m8, m15 = _build(torch, name8, cfg8, sd8), _build(torch, name15, cfg15, sd15)
    wav8 = torch.from_numpy(weights.make_waveforms(3000, seed=41)).cuda()
    wav15 = torch.from_numpy(weights.make_waveforms(300, seed=42)).cuda()
    want8, want15 = m8.forward_wav(wav8).clone(), m15.forward_wav(wav15).clone()
    s8, s15 = torch.cuda.Stream(), torch.cuda.Stream()
    out8 = [torch.empty_like(want8) for _ in range(4)]
    out15 = [torch.empty_like(want15) for _ in range(4)]
    torch.cuda.synchronize()
    for i in range(4):
        with torch.cuda.stream(s8):
            m8.forward_wav(wav8, out=out8[i])
        with torch.cuda.stream(s15):
            m15.forward_wav(wav15, out=out15[i])
    torch.cuda.synchronize()
    for i in range(4):
        assert torch.equal(out8[i], want8) and torch.equal(out15[i], want15), i
    # two cnn handles of the same model, each with its own side stream, unit counter and flag words, several chunks per call (conv_cols.hip: persistent workgroups of
    # BOTH launches share the CUs)
    _, namec, cfgc, sdc, _, _ = load_golden_model("model_cnn__cnn-trad-pool2.npz")
    ca, cb = _build(torch, namec, dict(cfgc, dtype="fp16"), sdc), _build(torch, namec, dict(cfgc, dtype="fp16"), sdc)
    fa = torch.from_numpy(weights.make_features(5000, seed=43)).cuda()
    fb = torch.from_numpy(weights.make_features(4100, seed=44)).cuda()
    wa, wb = ca(fa).clone(), cb(fb).clone()
    assert "two streams" in ca.plan_detail() and "conv_cols" in cb.plan_detail()
    oa = [torch.empty_like(wa) for _ in range(3)]
    ob = [torch.empty_like(wb) for _ in range(3)]
    torch.cuda.synchronize()
    for i in range(3):
        with torch.cuda.stream(s8):
            ca(fa, out=oa[i])
        with torch.cuda.stream(s15):
            cb(fb, out=ob[i])
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(oa[i], wa) and torch.equal(ob[i], wb), i


@pytest.mark.parametrize("fname,dtype", [("model_resnet__res15.npz", "f32"), ("model_resnet__res15.npz", "bf16"), ("model_resnet__res26.npz", "fp16"),
                                         ("model_cnn__cnn-trad-pool2.npz", "f32"), ("model_cnn__cnn-one-fstride4.npz", "fp16")])
def test_a_clips_logits_do_not_depend_on_its_neighbours(torch_cuda, fname, dtype):
    """Clips are independent units (BatchNorm in eval mode, reference run/test.py:21): on every plan -- the tiled kernels run over the
    flattened batch, workgroup tiles straddle clips -- a clip's logits are the same bits whatever batch it sits in, and wherever."""
    torch = torch_cuda
    from oracle import weights
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    model = _build(torch, name, dict(cfg, dtype=dtype), sd)
    x = torch.from_numpy(weights.make_features(53, seed=77)).cuda()
    full = model(x)
    perm = torch.randperm(53, generator=torch.Generator().manual_seed(5)).cuda()
    assert torch.equal(model(x[perm].contiguous()), full[perm])
    assert torch.equal(model(x[:7].contiguous()), full[:7])
    assert torch.equal(model(x[20:21].contiguous()), full[20:21])


def test_empty_batch_and_errors(torch_cuda):
    torch = torch_cuda
    from oracle import weights
    cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    model = _build(torch, "ResNet", cfg, weights.make_state_dict("ResNet", cfg, seed=7))
    assert model(torch.zeros(0, 101, 40, device="cuda")).shape == (0, 12)
    with pytest.raises(RuntimeError):
        model(torch.zeros(2, 101, 40))                              # CPU tensor: no CPU path
    with pytest.raises(ValueError):
        model(torch.zeros(2, 101, 39, device="cuda"))
    model.train()
    with pytest.raises(RuntimeError):
        model(torch.zeros(2, 101, 40, device="cuda"))


# ------------------------------------------------------------------ evaluation tail + evaluate()
def test_evaluate_matches_reference_evaluate(torch_cuda):
    torch = torch_cuda
    from honk2_amd.loss_function import ce_loss
    from honk2_amd.metric import Acc, PerClassAcc
    from honk2_amd.run.test import evaluate
    from oracle import weights
    with open(os.path.join(GOLDEN, "evaluate_res8.json")) as f:
        g = json.load(f)                                 # produced by the reference's run/test.py:evaluate
    cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    model = _build(torch, "ResNet", cfg, weights.make_state_dict("ResNet", cfg, seed=g["weights_seed"]))
    n, b = g["n_batches"], g["batch"]
    feats = weights.make_features(n * b, seed=g["seed_features"])
    labels = weights.make_labels(n * b, 12, seed=g["seed_labels"])
    loader = [(torch.from_numpy(feats[i * b:(i + 1) * b]), torch.from_numpy(labels[i * b:(i + 1) * b])) for i in range(n)]
    names = ["yes", "no", "up", "down", "left", "right", "on", "off", "stop", "go", "__unknown__", "__silence__"]
    res = evaluate(torch.device("cuda:0"), "golden", model, loader, ce_loss, {"Acc": Acc(), "PerClassAcc": PerClassAcc()},
                   dict(enumerate(names)), progress=False)
    assert abs(res["loss"] - g["result"]["loss"]) < 1e-4
    assert res["metric_Acc"] == g["result"]["metric_Acc"]
    assert res["metric_PerClassAcc"] == g["result"]["metric_PerClassAcc"]
    # generic (non-fused) tail: any other loss callable goes through the per-batch path and agrees
    res2 = evaluate(torch.device("cuda:0"), "golden", model, loader, lambda o, t: ce_loss(o, t),
                    {"Acc": Acc(), "PerClassAcc": PerClassAcc()}, dict(enumerate(names)), progress=False)
    assert abs(res2["loss"] - res["loss"]) < 1e-5 and res2["metric_PerClassAcc"] == res["metric_PerClassAcc"]


def test_entry_point_with_reference_checkpoint_and_odd_clip_length(torch_cuda, tmp_path):
    torch = torch_cuda
    import shutil
    from honk2_amd.run.test import build_model, main
    from honk2_amd.utils import load_checkpoint_state
    from oracle import frontend, models
    # evaluate_model_dir -> best_model.pt written by the reference's Workspace._save
    cfg = json.load(open(os.path.join(os.path.dirname(GOLDEN), "configs", "res8_synthetic.json")))
    cfg["model"]["config"] = {"pool": [4, 3], "n_feature_maps": 19, "n_layers": 6, "use_dilation": False}
    cfg["SyntheticKWSDataset"]["num_samples"] = 64
    shutil.copy(os.path.join(GOLDEN, "checkpoint_res8_narrow_best_model.pt"), tmp_path / "best_model.pt")
    cfg["evaluate_model_dir"] = str(tmp_path)
    res = main(cfg)
    assert np.isfinite(res["loss"]) and 0.0 <= res["metric_Acc"] <= 1.0
    # the weights that were evaluated are the checkpoint's: logits equal the reference golden for those weights
    tag, name, mcfg, sd, feats, z = load_golden_model("model_resnet__res8_narrow.npz")
    model = build_model(json.loads(json.dumps(cfg))).to("cuda:0").eval()
    model.load_state_dict(load_checkpoint_state(str(tmp_path / "best_model.pt"))[0])
    got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
    assert np.abs(got - z["logits"]).max() < LOGIT_TOL
    # res8 on clips that are not one second long: the fused kernel is specific to 101 frames, the layer-wise plan takes over
    from oracle import weights
    cfg8 = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    sd8 = weights.make_state_dict("ResNet", cfg8, seed=7)
    m8 = _build(torch, "ResNet", cfg8, sd8)
    wav = weights.make_waveforms(5, n_samples=12000 + 77, seed=9)[2:]
    y = m8.forward_wav(torch.from_numpy(wav).cuda()).cpu().numpy()
    assert m8.plan_name() == "resnet_tiled"
    want = models.forward_numpy("ResNet", cfg8, sd8, frontend.compute_mfccs_batch(wav, "f64"), np.float64)
    assert np.abs(y - want).max() < LOGIT_TOL and (y.argmax(1) == want.argmax(1)).all()


def test_streaming_windows_read_in_place(torch_cuda, monkeypatch):
    """kws_mfcc_windows / kws_forward_windows (reference dataset/dataset_utils.py:20-98: item i is stream[i*shift : i*shift +
    window]) against copying every window out and taking the batched path -- each window keeps its own reflect padding.
    With the shipped 1000 ms / 10 ms geometry 97 of a window's 101 frames are frames of the stream itself and are computed
    once (bit-identical to the batched path); the 2 + 2 edge frames come from a separate direct-DFT kernel and agree to the
    front end's tolerance.  With sharing switched off everything is bit-identical.  The loader's streaming shortcut must
    yield the same batches as item-by-item collation."""
    torch = torch_cuda
    import random
    from honk2_amd.data_loader import AudioDataLoader
    from honk2_amd.dataset import SyntheticStreamingDataset
    from honk2_amd.utils import AudioProcessor
    from oracle import frontend, models, weights
    cfg = dict(sample_rate=16000, target_class=["yes", "no", "up"], unknown_class=True, silence_class=True, type="dev",
               window_size_ms=1000, shift_size_ms=10, num_files=4, seed=3)
    random.seed(7)
    ds = SyntheticStreamingDataset(cfg)
    stream, window, shift, targets = ds.stream_view()
    n = len(ds)
    assert n == 300 and len(targets) == n
    gs = torch.from_numpy(stream).cuda()
    stacked = torch.from_numpy(np.stack([stream[i * shift:i * shift + window] for i in range(n)])).cuda()
    ap = AudioProcessor()
    f_ref = ap.compute_mfccs_batch(stacked)
    mcfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 5}
    sd = weights.make_state_dict("ResNet", mcfg, seed=2)
    model = _build(torch, "ResNet", mcfg, sd)
    l_ref = model.forward_wav(stacked)

    # shared-frame path
    f_win = ap.compute_mfccs_windows(gs, window, shift)
    assert f_win.shape == (n, 101, 40)
    assert torch.equal(f_win[:, 2:99], f_ref[:, 2:99])                       # the stream's own frames
    edge = [0, 1, 99, 100]
    # the 2 + 2 edge frames of EVERY window (window_edges_kernel) are held to the batched front end's own tiers against the float64
    # restatement (test_frontend_matches_oracle): 1e-3 on bands within 40 dB of the window's strongest band, 1e-2 within 60 dB
    wins = np.stack([stream[i * shift:i * shift + window] for i in range(n)])
    want = frontend.compute_mfccs_batch(wins, "f64")[:, edge]
    mel = frontend.mel_power(wins, "f64")                                    # (window, frame, band)
    top = np.maximum(mel.max(axis=(1, 2), keepdims=True), 1e-30)
    mel_e = mel[:, edge]
    for name, got_e in (("windows", f_win[:, edge].cpu().numpy()), ("batched", f_ref[:, edge].cpu().numpy())):
        err = np.abs(got_e - want)
        assert err[mel_e > 1e-4 * top].max() < 1e-3, (name, err[mel_e > 1e-4 * top].max())
        assert err[mel_e > 1e-6 * top].max() < 1e-2, (name, err[mel_e > 1e-6 * top].max())
    part = ap.compute_mfccs_windows(gs, window, shift, first=100, count=37)
    assert torch.equal(part[:, 2:99], f_ref[100:137, 2:99]) and torch.equal(part[:, edge], f_win[100:137][:, edge])   # a sub-range is the same bits
    l_win = model.forward_windows(gs, window, shift)
    assert (l_win - l_ref).abs().max() < LOGIT_TOL and torch.equal(l_win.argmax(1), l_ref.argmax(1))
    with pytest.raises(ValueError):
        model.forward_windows(gs, window, shift, first=n - 3, count=10)

    # every window on its own: bit-identical to the batched path (also what a shift that is no multiple of the hop gets)
    monkeypatch.setenv("KWS_WINDOWS_NO_SHARE", "1")
    assert torch.equal(ap.compute_mfccs_windows(gs, window, shift), f_ref)
    assert torch.equal(model.forward_windows(gs, window, shift), l_ref)
    monkeypatch.delenv("KWS_WINDOWS_NO_SHARE")
    odd = ap.compute_mfccs_windows(gs, window, 16 * 7)                        # 7 ms shift: no frame is shared
    odd_ref = ap.compute_mfccs_batch(torch.from_numpy(np.stack([stream[i * 112:i * 112 + window] for i in range(odd.shape[0])])).cuda())
    assert torch.equal(odd, odd_ref)

    # loader: streaming shortcut vs generic path (forced by asking for raw waveforms and running the front end by hand)
    fast = list(AudioDataLoader({"audio_preprocessing": "MFCCs", "batch_size": 128}, ds))
    slow = list(AudioDataLoader({"audio_preprocessing": "MFCCs", "batch_size": 128, "raw_waveforms": True}, ds))
    assert len(fast) == len(slow) == 3
    for (ff, ft), (sw, st) in zip(fast, slow):
        fr = ap.compute_mfccs_batch(sw)
        assert torch.equal(ft, st) and torch.equal(ff[:, 2:99], fr[:, 2:99]) and (ff - fr).abs().max() < 0.5
    assert torch.equal(torch.cat([t for _, t in fast]), torch.from_numpy(targets))


def test_data_loader_and_entry_point(torch_cuda, tmp_path):
    torch = torch_cuda
    from honk2_amd.run.test import main
    from oracle import frontend
    cfg = json.load(open(os.path.join(os.path.dirname(GOLDEN), "configs", "res8_synthetic.json")))
    cfg["SyntheticKWSDataset"]["num_samples"] = 96
    res = main(cfg)
    assert set(res) == {"loss", "metric_Acc", "metric_PerClassAcc"} and np.isfinite(res["loss"])
    from honk2_amd.dataset import DatasetType
    from honk2_amd.run.run_utils import init_data_loader
    loader = init_data_loader(cfg, DatasetType.TEST)
    feats, target = next(iter(loader))
    assert feats.shape == (32, 101, 40) and feats.is_cuda and target.shape == (32,)
    wav = np.stack([loader.dataset[i][0] for i in range(32)])
    assert np.abs(feats.cpu().numpy() - frontend.compute_mfccs_batch(wav, "f64")).max() < 1e-3


# ------------------------------------------------------------------ BASELINE batch sizes, every chunk loop value-checked
def _bulk_features(torch, n, seed, golden_feats):
    """n feature maps ~ N(0.65, 2.5) generated on the device; rows 0..len(golden)-1 are the golden fixture's inputs."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((n, 101, 40), generator=g, device="cuda") * 2.5 + 0.65
    x[:len(golden_feats)] = torch.from_numpy(golden_feats).cuda()
    return x


@pytest.mark.parametrize("fname,dtype,batch,tol", [("model_resnet__res15.npz", "bf16", 4096, 2e-2),
                                                    ("model_cnn__cnn-trad-pool2.npz", "fp16", 8192, 5e-3)])
def test_reduced_precision_configs_at_baseline_batch(torch_cuda, fname, dtype, batch, tol):
    """BASELINE configs[2] (res15 bf16, B = 4096) and configs[4] (cnn-trad-pool2 fp16, B = 8192) at their own batch: the
    layer-wise plans walk the batch in chunks of 1024 clips, so rows 0-5 must equal the golden-fixture run bit for bit
    (and sit within the dtype's tolerance of the reference), every clip must be independent of its place in the batch
    (permutation, a sub-batch straddling clip 1024, a ragged tail chunk) and the run must repeat exactly."""
    torch = torch_cuda
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    model = _build(torch, name, dict(cfg, dtype=dtype), sd)
    small = model(torch.from_numpy(feats).cuda())
    x = _bulk_features(torch, batch, 31, feats)
    full = model(x)
    assert full.shape == (batch, cfg["n_labels"]) and torch.isfinite(full).all()
    assert torch.equal(full[:len(feats)], small)                               # chunked run == the 6-clip fixture run
    want = z["logits"]
    assert np.abs(small.cpu().numpy() - want).max() < tol * max(1.0, float(np.abs(want).max()))
    assert torch.equal(model(x), full)                                         # deterministic
    g = torch.Generator(device="cuda").manual_seed(7)
    perm = torch.randperm(batch, device="cuda", generator=g)
    assert torch.equal(model(x[perm]), full[perm])                             # clips are independent units
    assert torch.equal(model(x[1000:1100]), full[1000:1100])                   # straddles the first chunk boundary
    assert torch.equal(model(x[3:batch - 517]), full[3:batch - 517])           # shifted chunks, ragged last chunk


@pytest.mark.parametrize("name,cfg_or_file,plan", [
    ("CNN", "model_cnn__cnn-trad-pool2.npz", "cnn_band"),
    ("CNN", "model_cnn__cnn-tstride4.npz", "cnn_band"),
    ("CNN", "model_cnn__cnn-tpool2.npz", "cnn_band"),
    ("CNN", "model_cnn__cnn-trad-pool2.npz", "layerwise"),
    ("CNN", "model_cnn__cnn-tstride4.npz", "layerwise"),
    ("ResNet", {"n_feature_maps": 30, "n_layers": 5, "use_dilation": True, "n_labels": 12}, "layerwise"),
    ("ResNet", {"n_feature_maps": 45, "n_layers": 4, "use_dilation": True, "n_labels": 12}, "resnet_tiled"),
])
def test_chunk_loops_against_the_oracle_at_1100_clips(torch_cuda, monkeypatch, name, cfg_or_file, plan):
    """1100 clips = one full 1024-clip chunk + a ragged second one through run_cnn (band plan: conv_band.hip behind a
    channels-last conv_0, and the generic kernels with KWS_CNN_BAND=0), run_resnet_layerwise (generic kernels, 30 feature
    maps) and run_resnet_tiled, every logit compared with the fp32 CPU oracle (reference model/cnn.py:79-107,
    model/resnet.py:38-60)."""
    torch = torch_cuda
    from oracle import models, weights
    if name == "CNN" and plan == "layerwise":
        monkeypatch.setenv("KWS_CNN_BAND", "0")
    if isinstance(cfg_or_file, str):
        _, _, cfg, _, _, _ = load_golden_model(cfg_or_file)
    else:
        cfg = cfg_or_file
    sd = weights.make_state_dict(name, cfg, seed=41)
    feats = weights.make_features(1100, seed=42)
    model = _build(torch, name, cfg, sd)
    got = model(torch.from_numpy(feats).cuda()).cpu().numpy()
    assert model.plan_name() == plan
    want = models.forward_torch(name, cfg, sd, feats).numpy()
    err = np.abs(got - want)
    assert err.max() < LOGIT_TOL, (cfg, err.max())
    assert err[1024:].max() < LOGIT_TOL and err[:1024].max() < LOGIT_TOL
    top = np.sort(want, axis=1)
    clear = (top[:, -1] - top[:, -2]) > 2 * err.max() + 1e-6
    assert (got.argmax(1) == want.argmax(1))[clear].all() and clear.mean() > 0.9


def test_evaluate_rejects_out_of_range_targets(torch_cuda):
    """A label outside [0, n_labels) must never be used as an index on the device (the reference's F.cross_entropy raises,
    loss_function.py:6-9): the fused tail skips the clip, counts it, and evaluate() raises after its one copy."""
    torch = torch_cuda
    from honk2_amd.loss_function import ce_loss
    from honk2_amd.metric import Acc, PerClassAcc
    from honk2_amd.run.test import evaluate
    from oracle import weights
    cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    model = _build(torch, "ResNet", cfg, weights.make_state_dict("ResNet", cfg, seed=3))
    feats = torch.from_numpy(weights.make_features(40, seed=4))
    good = torch.from_numpy(weights.make_labels(40, 12, seed=5))
    names = dict(enumerate("abcdefghijkl"))
    for bad_value in (12, -1, -100, 2 ** 40):
        bad = good.clone()
        bad[17] = bad_value
        with pytest.raises(IndexError):
            evaluate(torch.device("cuda:0"), "t", model, [(feats, bad)], ce_loss, {"Acc": Acc(), "PerClassAcc": PerClassAcc()},
                     names, progress=False)
    # the counters next to the stats block are untouched by the skipped clip, and the other 39 clips are counted
    eng = model.engine()
    logits = model(feats.cuda())
    guard = torch.full((64,), 7, dtype=torch.int64, device="cuda")
    stats = guard[8:8 + 27]
    stats.zero_()
    loss = torch.zeros(1, dtype=torch.float64, device="cuda")
    bad = good.clone()
    bad[17] = 12
    eng.eval_batch(logits, bad.cuda(), stats, loss)
    host = guard.cpu().tolist()
    assert host[:8] == [7] * 8 and host[35:] == [7] * 29
    assert host[8 + 1] == 39 and host[8 + 26] == 1 and sum(host[8 + 14:8 + 26]) == 39
    ok = evaluate(torch.device("cuda:0"), "t", model, [(feats, good)], ce_loss, {"Acc": Acc(), "PerClassAcc": PerClassAcc()},
                  names, progress=False)
    assert np.isfinite(ok["loss"])


def test_bench_two_ranks_rehearsal_matches_one_rank(torch_cuda, tmp_path):
    """bench.py's N > 1 leg (contiguous clip shards, all-gather of the logits in rank order; reference DataParallel gather,
    run/test.py:69-70) run as two fresh torchrun ranks sharing the one GPU over gloo (rehearsal knobs), against the
    single-rank run of the same global batch: same JSON contract, identical gathered logits."""
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # `python bench.py --gpus 2` AS TYPED (no torchrun in front): the script starts its two ranks itself, as fresh children, and relays rank 0's line
    env = dict(os.environ, KWS_BENCH_BACKEND="gloo", KWS_BENCH_ONE_DEVICE="1", KWS_BENCH_DUMP=str(tmp_path / "n2.npy"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2048", "--prewarm-ms", "0"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert two.returncode == 0, two.stderr[-2000:]
    lines2 = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines2) == 1, lines2                                     # ONE JSON line, from rank 0
    line2 = json.loads(lines2[-1])
    assert line2["n_gpus"] == 2 and line2["config"]["clips_per_gpu"] == 1024 and line2["config"]["global_batch"] == 2048
    coll2 = line2["collective"]
    assert coll2["world_size"] == 2 and coll2["clips_per_gpu"] == [1024, 1024] and coll2["backend"] == "gloo"
    assert [d["rank"] for d in coll2["devices"]] == [0, 1] and coll2["gathered_equals_local_shard_on_rank_0"]
    # more ranks than GPUs without the rehearsal knob: every rank but the first has no device -- the run must fail loudly, not shrink
    env_bad = dict(env)
    env_bad.pop("KWS_BENCH_ONE_DEVICE")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2048", "--prewarm-ms", "0"],
                         env=env_bad, capture_output=True, text=True, timeout=600, cwd=root)
    if torch_cuda.cuda.device_count() < 2:
        assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]
    # the explicit torchrun form the driver documents stays valid
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                         "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2048", "--prewarm-ms", "0"],
                        env=dict(env, KWS_BENCH_DUMP=str(tmp_path / "n2t.npy")), capture_output=True, text=True, timeout=600, cwd=root)
    assert tr.returncode == 0, tr.stderr[-2000:]
    assert np.array_equal(np.load(tmp_path / "n2t.npy"), np.load(tmp_path / "n2.npy"))
    assert line2["steps"] == 2 and line2["warmup"] == 1 and line2["value"] > 0 and line2["scaling"] == "strong"
    assert "cpu_baseline" not in line2 and line2["roofline"]["launches"] == 2 and line2["prewarm"]["steps"] == 0
    env1 = dict(os.environ, KWS_BENCH_DUMP=str(tmp_path / "n1.npy"))
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--batch", "2048", "--no-cpu-baseline", "--no-secondary"], env=env1, capture_output=True, text=True, timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    line1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert line1["n_gpus"] == 1 and line1["config"]["clips_per_gpu"] == 2048
    assert 0 < line1["prewarm"]["steps"] <= 400 and line1["roofline"]["launches"] == 2     # the untimed clock-governor pre-warm is not in the timed region
    a, b = np.load(tmp_path / "n1.npy"), np.load(tmp_path / "n2.npy")
    assert a.shape == b.shape == (2048, 12) and np.array_equal(a, b)
    # the RCCL leg gathers on a side stream while the next step computes (two buffers in rotation): the same stream / event logic
    # with a device copy standing in for the collective must hand back the last step's logits
    env3 = dict(os.environ, KWS_BENCH_DUMP=str(tmp_path / "n1g.npy"), KWS_BENCH_SELF_GATHER="1")
    for steps in ("2", "3"):
        sg = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", steps, "--warmup", "1",
                             "--batch", "2048", "--no-cpu-baseline", "--no-secondary"], env=env3, capture_output=True, text=True, timeout=600, cwd=root)
        assert sg.returncode == 0, sg.stderr[-2000:]
        assert np.array_equal(np.load(tmp_path / "n1g.npy"), a)
    # RCCL itself, on the hardware this box has: the distributed leg as a fresh torchrun child with a ONE-rank nccl process group
    # (KWS_FORCE_DIST): init_process_group("nccl"), the side-stream all_gather_into_tensor with its event hand-offs, barrier and the
    # MAX all-reduce of the step time all execute on RCCL; the gathered logits must be the local ones, via the overlapped path
    # (not the compute-stream fallback).
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env4 = dict(os.environ, KWS_FORCE_DIST="1", KWS_BENCH_DUMP=str(tmp_path / "n1r.npy"))
    env4.pop("KWS_BENCH_BACKEND", None)
    for steps in ("2", "3"):
        rc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                             "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                             "--gpus", "1", "--steps", steps, "--warmup", "1", "--batch", "2048", "--no-cpu-baseline", "--no-secondary",
                             "--no-shard", "--no-h2d"], env=env4, capture_output=True, text=True, timeout=600, cwd=root)
        assert rc.returncode == 0, rc.stderr[-2000:]
        liner = json.loads([ln for ln in rc.stdout.splitlines() if ln.startswith("{")][-1])
        coll = liner["collective"]
        assert coll["backend"] == "nccl" and coll["world_size"] == 1 and coll["forced_one_rank_group"] and coll["rccl_version"]
        assert coll["overlapped_on_side_stream"] and coll["overlap_fallback"] is None, coll
        assert np.array_equal(np.load(tmp_path / "n1r.npy"), a)


def test_entry_point_reduces_over_rccl_with_a_one_rank_group(torch_cuda, tmp_path):
    """run/test.py:main under torchrun with a forced one-rank nccl process group: shard_loader, evaluate_partial and reduce_results'
    all-reduce run on RCCL (reference: DataParallel, run/test.py:69-70); the result must equal the plain single-process evaluation."""
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = os.path.join(root, "tests", "configs", "res8_synthetic.json")
    script = tmp_path / "run_eval.py"
    script.write_text(
        "import json, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "import torch.distributed as dist\n"
        "from honk2_amd.run import test as t\n"
        "from honk2_amd.utils import load_json\n"
        "res = t.main(load_json(sys.argv[1]))\n"
        "info = {'backend': dist.get_backend() if dist.is_initialized() else None}\n"
        "json.dump({'res': res, 'info': info}, open(sys.argv[2], 'w'))\n")
    plain = subprocess.run([sys.executable, str(script), cfg, str(tmp_path / "plain.json")], capture_output=True, text=True, timeout=600, cwd=root)
    assert plain.returncode == 0, plain.stderr[-2000:]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    forced = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                             "--master-port", str(port), str(script), cfg, str(tmp_path / "forced.json")],
                            env=dict(os.environ, KWS_FORCE_DIST="1"), capture_output=True, text=True, timeout=600, cwd=root)
    assert forced.returncode == 0, forced.stderr[-2000:]
    a, b = json.load(open(tmp_path / "plain.json")), json.load(open(tmp_path / "forced.json"))
    assert a["info"]["backend"] is None and b["info"]["backend"] == "nccl"
    assert a["res"]["metric_Acc"] == b["res"]["metric_Acc"] and a["res"].get("metric_PerClassAcc") == b["res"].get("metric_PerClassAcc")
    assert abs(a["res"]["loss"] - b["res"]["loss"]) <= 1e-12 * max(1.0, abs(a["res"]["loss"]))
    # `python -m honk2_amd.run.test --config X` AS TYPED with more than one rank wanted (num_gpu > 1 on a multi-GPU node; here the rehearsal
    # knobs: two ranks on the one GPU over gloo): the command starts its ranks itself and rank 0 prints the single-process result
    import ast
    env2 = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env2.update(KWS_EVAL_RANKS="2", KWS_BENCH_BACKEND="gloo", KWS_BENCH_ONE_DEVICE="1")
    typed = subprocess.run([sys.executable, "-m", "honk2_amd.run.test", "--config", cfg], env=env2, capture_output=True, text=True, timeout=600, cwd=root)
    assert typed.returncode == 0, typed.stderr[-2000:]
    assert typed.stdout.count("Test results") == 1 and "(per rank, 2 ranks)" in typed.stdout
    c = ast.literal_eval(typed.stdout.split("Test results", 1)[1].strip())
    assert c["metric_Acc"] == a["res"]["metric_Acc"] and c.get("metric_PerClassAcc") == a["res"].get("metric_PerClassAcc")
    assert abs(c["loss"] - a["res"]["loss"]) <= 1e-9 * max(1.0, abs(a["res"]["loss"]))


def test_fp16_cells_between_conv_1_and_the_linear_change_no_bit(torch_cuda, monkeypatch):
    """`fp16` dtype on the band plan (reference model/cnn.py:79-107 as a `.half()` model): the first Linear forms single-term fp16 products, i.e. it rounds
    its input to fp16 itself -- so conv_1 may store exactly that fp16 value (half the bytes written and read).  Logits must equal the fp32-cell form bit
    for bit, on the two-conv models that take the band plan, over a batch of more than one chunk."""
    torch = torch_cuda
    from oracle import weights
    for fname, plan in (("model_cnn__cnn-trad-pool2.npz", "cnn_band"), ("model_cnn__cnn-tstride4.npz", "cnn_band"), ("model_cnn__cnn-tpool2.npz", "cnn_band"),
                        ("model_cnn__cnn-one-fpool3.npz", "cnn_in1"), ("model_cnn__cnn-one-fstride4.npz", "cnn_in1")):   # (single-conv models: conv_0's cells feed the Linear)
        tag, name, cfg, sd, feats, z = load_golden_model(fname)
        x = torch.from_numpy(weights.make_features(4500, seed=21)).cuda()      # (more than one chunk on every model: chunks hold 1 024 - 4 096 clips)
        outs = {}
        for knob in ("1", "0"):
            monkeypatch.setenv("KWS_CNN_LIN_F16", knob)
            m = _build(torch, name, dict(cfg, dtype="fp16"), sd)
            outs[knob] = m(x)
            assert m.plan_name() == plan, (fname, m.plan_name())
        assert torch.isfinite(outs["1"]).all() and torch.equal(outs["1"], outs["0"]), (fname, float((outs["1"] - outs["0"]).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("fname,dtype", [("model_cnn__cnn-trad-pool2.npz", "fp16"), ("model_cnn__cnn-trad-pool2.npz", "f32"),
                                         ("model_cnn__cnn-one-fstride4.npz", "fp16"), ("model_cnn__cnn-tstride8.npz", "f32")])
def test_cnn_two_stream_chunks_change_no_bit(torch_cuda, monkeypatch, fname, dtype):
    """cnn plans, calls of more than one chunk (reference model/cnn.py:79-107 over a batch): a chunk's Linears, split-K reduces and the range guard's gated
    second pass run on a stream of the handle while the caller's stream computes the next chunk's convolutions (run_cnn, two buffers for what the Linear
    reads, three flag words).  Same kernels on the same operands, so 13 000 clips -- four chunks or more, the SECOND one driven out of fp16's range so
    that its recomputation overlaps the third chunk's first pass -- must give the one-stream form's logits bit for bit, eagerly, from a stream that is not
    the default one, and replayed from a captured graph."""
    torch = torch_cuda
    from oracle import weights
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    n = 13000
    monkeypatch.setenv("KWS_CNN_STREAMS", "0")
    one = _build(torch, name, dict(cfg, dtype=dtype), sd)
    c = one.chunk_clips(n)                            # 1 024 - 4 096 clips by the model's footprint: at least four chunks here
    assert 4 * c <= n + 3 * 64, (c, n)
    f = weights.make_features(n, seed=41)
    f[c:2 * c] *= 40000.0
    x = torch.from_numpy(f).cuda()
    want = one(x).clone()
    monkeypatch.delenv("KWS_CNN_STREAMS")
    two = _build(torch, name, dict(cfg, dtype=dtype), sd)
    for _ in range(3):
        got = two(x)
        assert torch.isfinite(got).all() and torch.equal(got, want), float((got - want).abs().max())
    assert two.plan_name() == one.plan_name()
    big = float(want[c:2 * c].abs().max())
    assert big > 3e3                                               # the driven chunk really was recomputed
    if dtype == "f32":                                             # ... to fp32 accuracy
        from oracle import models
        ref = models.forward_torch(name, cfg, sd, f[c:c + 512]).numpy()
        assert np.abs(want[c:c + 512].cpu().numpy() - ref).max() < 3e-6 * big
    static_in = x.clone()
    out = torch.empty((n, cfg["n_labels"]), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        two(static_in, out=out)
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        with torch.cuda.graph(graph, stream=side):
            two(static_in, out=out)
    torch.cuda.current_stream().wait_stream(side)
    x2 = torch.from_numpy(weights.make_features(n, seed=42)).cuda()
    want2 = one(x2).clone()
    for src, w in ((x2, want2), (x, want), (x2, want2)):
        static_in.copy_(src)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, w), float((out - w).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("fname", ["model_cnn__cnn-trad-pool2.npz", "model_cnn__cnn-trad-fpool3.npz"])
def test_column_tile_conv_1_agrees_with_the_band_kernel(torch_cuda, monkeypatch, fname):
    """`fp16` dtype, conv_1 of the two shipped configs it fits (reference model/cnn.py:82-92): conv_cols.hip forms the same single-term fp16 products as
    conv_band.hip and sums them in another order (kernel row, channel quad, kernel column instead of tap, channel block) -- so over 4 500 clips (three chunks or
    more, two streams) the two must agree to the rounding of the accumulations and of the fp16 cells behind them, far inside the fp16 bar both are held to against the oracle, and a clip's
    logits must not depend on its neighbours or on which workgroup drew its bands."""
    torch = torch_cuda
    from oracle import models, weights
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    f = weights.make_features(4500, seed=51)
    x = torch.from_numpy(f).cuda()
    cols = _build(torch, name, dict(cfg, dtype="fp16"), sd)
    got = cols(x)
    assert "conv_cols" in cols.plan_detail() and "two streams" in cols.plan_detail(), cols.plan_detail()
    monkeypatch.setenv("KWS_CNN_COLS", "0")
    band = _build(torch, name, dict(cfg, dtype="fp16"), sd)
    ref = band(x)
    assert "conv_band" in band.plan_detail(), band.plan_detail()
    scale = max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) < 1e-3 * scale, float((got - ref).abs().max())      # (the fp16 cells behind conv_1 round the two sums an ulp apart here and there: 2.4e-4 seen over 4 500 clips)
    want = models.forward_torch(name, cfg, sd, f[:300]).numpy()
    assert np.abs(got[:300].cpu().numpy() - want).max() < 5e-3 * max(1.0, float(np.abs(want).max()))
    for _ in range(2):                                   # the unit counter hands bands out in a different order every launch
        assert torch.equal(cols(x), got)
    assert torch.equal(cols(x[37:300]), got[37:300])     # ... and a clip's logits do not depend on where it sits in the batch
