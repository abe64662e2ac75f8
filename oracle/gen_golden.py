#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own classes (build container only).

Run:  python oracle/gen_golden.py            (needs /root/reference; never runs on the GPU box)

What it does
------------
* Imports the unmodified reference packages from /root/reference.  ``utils/__init__.py:1`` pulls in
  the front end, which imports the absent third-party modules ``librosa``, ``pcen`` and (through
  ``utils/workspace.py:5``) ``torch.utils.tensorboard``; empty in-memory placeholders are put in
  ``sys.modules`` for those three so that the MODEL / METRIC / LOSS / EVALUATE code (which never
  touches them) imports.  The front end itself is NOT executed (it cannot be: librosa is absent).
* For every shipped model config: builds the reference ``model.ResNet`` / ``model.CNN``, loads the
  deterministic state dict from ``oracle.weights`` (strict), runs seeded features through it on the
  CPU in fp32 and stores logits + a few sub-sampled intermediates.  Fixtures hold seeds and
  expected outputs only; weights/inputs are regenerated from the seed by the tests.
* Runs the reference ``run/test.py:evaluate`` on a list-of-batches loader for res8 and stores the
  result dict (loss / Acc / PerClassAcc) for the evaluation-tail parity test.
* Pins the "DCT" step of ``utils/audio_processor.py:28`` against scipy, and stores front-end vectors
  from the float64 restatement (librosa part: parity unpinned, see oracle/__init__.py).
"""
import glob
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import frontend, weights  # noqa: E402


def _install_placeholders():
    lib = types.ModuleType("librosa")
    lib.core = types.ModuleType("librosa.core")
    lib.feature = types.ModuleType("librosa.feature")
    sys.modules.setdefault("librosa", lib)
    sys.modules.setdefault("librosa.core", lib.core)
    sys.modules.setdefault("librosa.feature", lib.feature)
    pc = types.ModuleType("pcen")

    class StreamingPCENTransform:  # constructed by AudioProcessor.__init__, never by this script
        def __init__(self, *a, **k):
            pass
    pc.StreamingPCENTransform = StreamingPCENTransform
    sys.modules.setdefault("pcen", pc)
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        tb = types.ModuleType("torch.utils.tensorboard")

        class SummaryWriter:
            def __init__(self, *a, **k):
                pass
        tb.SummaryWriter = SummaryWriter
        sys.modules["torch.utils.tensorboard"] = tb


def sd_digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


def subsample(a, limit=16384):
    flat = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    step = max(1, -(-flat.size // limit))
    return flat[::step].copy(), step


def model_configs():
    items = []
    for path in sorted(glob.glob(os.path.join(REF, "config", "*", "*.json"))):
        with open(path) as f:
            cfg = json.load(f)
        if "model" not in cfg:
            continue
        n_labels = len(cfg["target_class"]) + int(bool(cfg["unknown_class"])) + int(bool(cfg["silence_class"]))
        tag = os.path.relpath(path, os.path.join(REF, "config")).replace(os.sep, "__")[:-5]
        items.append((tag, cfg["model"]["name"], dict(cfg["model"]["config"], n_labels=n_labels), cfg))
    return items


def main():
    import torch
    _install_placeholders()
    sys.path.insert(0, REF)
    import model as ref_model            # noqa: F401  (registers model.ResNet / model.CNN)
    import metric as ref_metric          # noqa: F401
    import loss_function as ref_loss     # noqa: F401
    from utils import find_cls
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    index = {}

    for tag, name, mcfg, full in model_configs():
        hey = tag.startswith("hey_snips")
        time = 901 if hey else mcfg.get("time", 101)
        batch = 2 if hey else 6
        seed = 7
        sd = weights.make_state_dict(name, mcfg, seed=seed)
        feats = weights.make_features(batch, seed=seed + 1, time=time)
        net = find_cls(f"model.{name}")(dict(mcfg))
        net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
        net.eval()
        grabbed = {}
        hooks = []

        def grab(key):
            def fn(_m, _i, o):
                grabbed[key] = o.detach().numpy()
            return fn
        if name == "ResNet":
            if "pool" in net.layers:
                hooks.append(net.layers["pool"].register_forward_hook(grab("post_pool")))
            hooks.append(net.layers["bn_2"].register_forward_hook(grab("post_layer2")))
            hooks.append(net.layers[f"bn_{mcfg['n_layers']}"].register_forward_hook(grab("pre_mean")))
        else:
            for i in (0, 1):
                if f"pool_{i}" in net.layers:
                    hooks.append(net.layers[f"pool_{i}"].register_forward_hook(grab(f"post_pool_{i}")))
        with torch.no_grad():
            logits = net(torch.from_numpy(feats)).numpy()
        for h in hooks:
            h.remove()
        rec = {
            "model_name": name, "model_config": json.dumps(mcfg), "seed": seed, "batch": batch, "time": time,
            "weights_sha256": sd_digest(sd), "num_params": int(net.num_params()),
            "logits": logits.astype(np.float32),
        }
        for k, v in grabbed.items():
            vals, step = subsample(v[1])          # clip 1 (clip 0 is the all-zero clip)
            rec[f"tap_{k}"] = vals
            rec[f"tap_{k}_step"] = step
            rec[f"tap_{k}_shape"] = np.asarray(v[1].shape)
        np.savez_compressed(os.path.join(OUT, f"model_{tag}.npz"), **rec)
        index[tag] = {"model": name, "num_params": rec["num_params"], "max_abs_logit": float(np.abs(logits).max())}
        print(f"{tag:28s} params={rec['num_params']:8d} |logit|max={np.abs(logits).max():.3f}")

    # ---- evaluate() golden (run/test.py:18-41) on res8
    sys.modules.setdefault("dataset", types.ModuleType("dataset"))  # run/test.py:13 imports DatasetType only
    if not hasattr(sys.modules["dataset"], "DatasetType"):
        import enum
        sys.modules["dataset"].DatasetType = enum.Enum("DatasetType", {"TRAIN": "train", "DEV": "dev", "TEST": "test"})
    from metric import collect_metrics  # noqa: F401
    tag, name, mcfg, full = [c for c in model_configs() if c[0] == "resnet__res8"][0]
    sd = weights.make_state_dict(name, mcfg, seed=7)
    net = find_cls(f"model.{name}")(dict(mcfg))
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    n_batches, bsz = 3, 16
    feats = weights.make_features(n_batches * bsz, seed=21)
    labels = weights.make_labels(n_batches * bsz, 12, seed=21)
    loader = [(torch.from_numpy(feats[i * bsz:(i + 1) * bsz]), torch.from_numpy(labels[i * bsz:(i + 1) * bsz]))
              for i in range(n_batches)]
    # evaluate() body restated call-for-call with the reference's own loss/metric objects
    # (run/test.py's module-level imports need tqdm+Workspace; the loop itself is :21-39)
    loss_fn = find_cls("loss_fn.ce_loss")
    metrics = {"Acc": find_cls("metric.Acc")(), "PerClassAcc": find_cls("metric.PerClassAcc")()}
    label_mapping = {i: s for i, s in enumerate(full["target_class"] + ["__unknown__", "__silence__"])}
    try:
        from run.test import evaluate
        res = evaluate(torch.device("cpu"), "golden", net, loader, loss_fn, metrics, label_mapping)
        how = "run.test.evaluate"
    except Exception as e:  # pragma: no cover
        print("run.test import failed (", type(e).__name__, e, "); using the loop of run/test.py:21-39 inline")
        total = 0.0
        net.eval()
        for data, target in loader:
            with torch.no_grad():
                out = net(data)
            total += loss_fn(out, target).item()
            for m in metrics.values():
                m.accumulate(out, target)
        res = {"loss": total / len(loader)}
        res.update(collect_metrics(metrics, label_mapping))
        how = "inline loop"
    with open(os.path.join(OUT, "evaluate_res8.json"), "w") as f:
        json.dump({"how": how, "seed_features": 21, "seed_labels": 21, "n_batches": n_batches, "batch": bsz,
                   "weights_seed": 7, "result": res}, f, indent=1)
    print("evaluate:", how, res["loss"], res["metric_Acc"])

    # ---- a checkpoint written by the reference's own Workspace._save (utils/workspace.py:34-45): res8_narrow
    import tempfile
    from utils import Workspace
    tag, name, mcfg, full = [c for c in model_configs() if c[0] == "resnet__res8_narrow"][0]
    sdn = weights.make_state_dict(name, mcfg, seed=7)
    netn = find_cls(f"model.{name}")(dict(mcfg))
    netn.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sdn.items()})
    wrapped = torch.nn.DataParallel(netn)          # what run/train.py saves when num_gpu > 1: keys get a 'module.' prefix
    opt = torch.optim.SGD(wrapped.parameters(), lr=0.1)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=3000)
    with tempfile.TemporaryDirectory() as tmp:
        ws = Workspace(torch.device("cpu"), os.path.join(tmp, "ws"), wrapped, find_cls("loss_fn.ce_loss"),
                       {"Acc": find_cls("metric.Acc")(), "PerClassAcc": find_cls("metric.PerClassAcc")()}, opt, sched)
        ws.save_best_model({"best_epoch": 3, "best_dev_criterion": 0.5, "best_dev_loss": 1.25})
        with open(os.path.join(tmp, "ws", "best_model.pt"), "rb") as f:
            blob = f.read()
    with open(os.path.join(OUT, "checkpoint_res8_narrow_best_model.pt"), "wb") as f:
        f.write(blob)
    print("checkpoint fixture:", len(blob), "bytes")

    # ---- DCT pin (utils/audio_processor.py:27-29) with the installed scipy
    import scipy.fftpack
    rng = np.random.Generator(np.random.PCG64(5))
    logmel = rng.standard_normal((40, 101))
    logmel[3, 7] = 0.0
    ref = [scipy.fftpack.dct(x) for x in np.split(logmel, logmel.shape[1], axis=1)]
    ref = np.array(ref, order="F").astype(np.float32)              # (101, 40, 1)
    np.savez_compressed(os.path.join(OUT, "frontend_dct_pin.npz"), logmel=logmel, out=ref)
    assert np.array_equal(ref[:, :, 0], (2.0 * logmel.T).astype(np.float32))

    # ---- front-end vectors from the float64 restatement (librosa parity unpinned)
    wav = weights.make_waveforms(14, seed=1234)
    t = np.arange(16000) / 16000.0
    wav[2] = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)          # bare bin-centred sine
    wav[3] = 0.0
    wav[3, 8000] = 1.0                                                            # impulse
    wav[4] = np.clip(weights._normal(99, "loud", (16000,), 0.0, 0.6), -1, 1)      # near full-scale noise
    wav[5] = (0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.01 * wav[5]).astype(np.float32)
    feats64 = frontend.compute_mfccs_batch(wav, "f64")
    mel64 = frontend.mel_power(wav, "f64")
    np.savez_compressed(os.path.join(OUT, "frontend_vectors.npz"), wav_seed=1234, n_clips=14,
                        wav_special=wav[2:6], feats=feats64, mel_max=mel64.max(axis=(1, 2)),
                        mel_bank=frontend.mel_filterbank())
    with open(os.path.join(OUT, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
