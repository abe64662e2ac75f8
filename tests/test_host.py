"""CPU: host logic (registry, config plumbing, shapes, C-ABI loading) -- no compute calls."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import honk2_amd
from conftest import GOLDEN, ROOT, golden_model_files, load_golden_model
from honk2_amd import _lib
from honk2_amd.utils import (calculate_conv_output_size, calculate_pool_output_size, find_cls, register_cls)
from honk2_amd.utils.trie import Trie


def test_registry_keys_and_trie_semantics():
    for key in ("model.ResNet", "model.CNN", "data_loader.AudioDataLoader", "metric.Acc", "metric.PerClassAcc",
                "loss_fn.ce_loss", "loss_fn.nll_loss", "dataset.SyntheticKWSDataset"):
        assert find_cls(key) is not None, key
    assert find_cls("model.Nope") is None and find_cls("model.Nope", 7) == 7
    t = Trie()
    t.add("a.b", 1)
    t.add("a.b.c", 2)
    assert (t.get("a.b"), t.get("a.b.c"), t.get("a.x", "d"), t.get("a")) == (1, 2, "d", None)
    assert t.count == 2 - 3            # reference quirk: every successful walk decrements (utils/trie.py:31)

    @register_cls("model.Tmp")
    class Tmp:
        pass
    assert find_cls("model.Tmp") is Tmp


def test_install_into_foreign_registry():
    seen = {}

    def foreign_register(key):
        def deco(obj):
            seen[key] = obj
            return obj
        return deco
    honk2_amd.install_into(foreign_register)
    assert seen["model.ResNet"] is find_cls("model.ResNet") and "data_loader.AudioDataLoader" in seen


def test_size_calculators_match_conv_arithmetic():
    assert calculate_conv_output_size([101, 40], (20, 8), stride=(1, 1)) == [82, 33]
    assert calculate_pool_output_size([82, 33], (2, 2)) == [41, 16]
    assert calculate_conv_output_size([101, 40], (16, 8), stride=(8, 1)) == [11, 33]
    assert calculate_conv_output_size([25, 13], (3, 3), padding=1) == [25, 13]
    assert calculate_conv_output_size([101, 40], (3, 3), padding=16, dilation=16) == [101, 40]
    assert calculate_pool_output_size([101, 40], (4, 3)) == [25, 13]


@pytest.mark.parametrize("fname", golden_model_files())
def test_models_build_with_reference_state_dict_layout(fname):
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    model = find_cls(f"model.{name}")(dict(cfg))
    assert model.num_params() == int(z["num_params"]) == model.num_trainable_params()
    own = model.state_dict()
    assert list(own.keys()) == list(sd.keys())
    for k, v in sd.items():
        assert tuple(own[k].shape) == tuple(np.shape(v)), k
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    with pytest.raises(RuntimeError):          # no GPU here and no CPU fallback, by design
        model(torch.from_numpy(feats))


def test_default_init_follows_torch_layer_init():
    torch.manual_seed(0)
    m = find_cls("model.ResNet")({"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12})
    torch.manual_seed(0)
    ref = [torch.nn.Conv2d(1, 45, 3, padding=1, bias=False)]
    for _ in range(6):
        ref.append(torch.nn.Conv2d(45, 45, 3, padding=1, bias=False))
        torch.nn.BatchNorm2d(45, affine=False)
    lin = torch.nn.Linear(45, 12)
    for i, conv in enumerate(ref):
        assert torch.equal(m.layers[f"conv_{i}"].weight, conv.weight)
    assert torch.equal(m.layers["output"].weight, lin.weight) and torch.equal(m.layers["output"].bias, lin.bias)


def test_c_abi_library_loads_and_exports_every_declared_symbol(lib_built):
    header = open(os.path.join(ROOT, "include", "kws.h")).read()
    declared = set(re.findall(r"\b(kws_[a-z_0-9]+)\s*\(", header))
    declared -= {"kws_handle", "kws_model_desc", "kws_conv_desc"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for sym in declared:
        assert hasattr(lib_built, sym), sym
    assert lib_built.kws_abi_version() == 2
    assert ctypes.sizeof(_lib.ModelDesc) == 4 * (11 + 1 + 10 + 4 + 3 + 4) + 8
    # without a GPU kws_create must fail loudly, not fall back
    if not torch.cuda.is_available():
        h = ctypes.c_void_p()
        d = _lib.make_desc(_lib.KWS_MODEL_NONE)
        rc = lib_built.kws_create(ctypes.byref(d), ctypes.byref(h))
        assert rc != 0 and b"no CPU fallback" in lib_built.kws_last_error()
        with pytest.raises(RuntimeError):
            _lib.Engine(d)


def test_no_exception_crosses_the_c_abi(lib_built, lib_experiments, monkeypatch):
    """include/kws.h: "never throws across the ABI".  In the EXPERIMENTS build KWS_TEST_THROW makes kws_create's body throw (std::bad_alloc
    or a std::runtime_error) in front of the device check, so this runs without a GPU: the guard around every entry point has to turn it
    into a code + kws_last_error(), not into std::terminate under ctypes.  The PRODUCT build has no such hook: with the variable set its
    kws_create goes on to the device check."""
    d = _lib.make_desc(_lib.KWS_MODEL_NONE)
    h = ctypes.c_void_p()
    monkeypatch.setenv("KWS_TEST_THROW", "bad_alloc")
    assert lib_experiments.kws_create(ctypes.byref(d), ctypes.byref(h)) == _lib.KWS_ENOMEM and not h.value
    assert b"out of host memory" in lib_experiments.kws_last_error()
    monkeypatch.setenv("KWS_TEST_THROW", "synthetic failure")
    assert lib_experiments.kws_create(ctypes.byref(d), ctypes.byref(h)) == _lib.KWS_EINVAL and not h.value
    assert b"synthetic failure" in lib_experiments.kws_last_error()
    rc = lib_built.kws_create(ctypes.byref(d), ctypes.byref(h))
    assert b"synthetic failure" not in lib_built.kws_last_error()
    if rc == 0:                                   # (a GPU is present: the handle is real)
        lib_built.kws_destroy(h)
    else:
        assert rc == _lib.KWS_EHIP and b"no HIP device" in lib_built.kws_last_error()


EXPERIMENT_SWITCHES = ("KWS_R8_DEBUG", "KWS_T3_DEBUG", "KWS_T3_TIMING", "KWS_T3_TIMING_LAYER", "KWS_BAND_TIMING", "KWS_TEST_THROW", "KWS_N_CU",
                       "KWS_R8_WGS_PER_CU", "KWS_R8_GRID", "KWS_FE_WGS_PER_CU", "KWS_TILED_CHUNK", "KWS_CNN_CHUNK", "KWS_KSPLIT_MIN_STEPS",
                       "KWS_T3_PAIR_WGS3", "KWS_T3_TRIPLE_CFG")


def test_product_library_has_no_experiment_switches(lib_built, lib_experiments):
    """The shipped libkws_hip.so must not compute wrong results, synchronise inside a compute call or throw on purpose because an
    environment variable is set ("no internal sync"): the names of the experiment switches do not even occur in it.
    They do occur in the EXPERIMENTS=1 build, which is what tools/ measures with."""
    prod = open(_lib.LIB_PATH, "rb").read()
    exp = open(_lib.EXP_LIB_PATH, "rb").read()
    for name in EXPERIMENT_SWITCHES:
        assert name.encode() not in prod, name
        assert name.encode() in exp, name
    # the implementation selectors that remain choose between implementations that are each parity-tested (include/kws.h lists them)
    for name in ("KWS_RES8_IMPL", "KWS_FRONTEND_IMPL", "KWS_LAYERWISE_IMPL"):
        assert name.encode() in prod, name


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "honk2_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_data_loader_collate_and_config_defaults():
    from honk2_amd.dataset import DatasetType, SyntheticKWSDataset
    from honk2_amd.run.run_utils import init_data_loader, merge_configs
    cfg = json.load(open(os.path.join(ROOT, "tests", "configs", "res8_synthetic.json")))
    loader = init_data_loader(cfg, DatasetType.TEST)      # reference configs lack shuffle/num_workers: defaults apply
    assert loader.batch_size == 32 and isinstance(loader.dataset, SyntheticKWSDataset)
    assert len(loader.dataset.label_mapping) == 12 and loader.dataset.label_mapping[11] == "__silence__"
    wav, target = loader.collate_fn([loader.dataset[i] for i in range(5)])
    assert wav.shape == (5, 16000) and wav.dtype == torch.float32 and target.dtype == torch.int64
    sil = [i for i in range(len(loader.dataset)) if loader.dataset.labels[i] == 11][0]
    assert not loader.dataset[sil][0].any()               # silence class = exact zeros (dataset/gsc_dataset.py:165-166)
    ragged = loader.collate_fn([(np.ones(10, np.float32), 1), (np.ones(7, np.float32), 2)])[0]
    assert ragged.shape == (2, 10) and ragged[1, 7:].sum() == 0
    base = {"a": {"x": 1}, "b": 2}
    merged = merge_configs(base, {"a": {"y": 3}})
    assert merged == {"a": {"y": 3}, "b": 2} and base["a"] == {"x": 1}      # shallow override, base untouched
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            next(iter(loader))


def test_metrics_and_collect_follow_reference_semantics():
    from honk2_amd.metric import Acc, PerClassAcc, collect_metrics
    out = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1], [0.2, 0.2, 0.6], [0.5, 0.5, 0.0]])
    tgt = torch.tensor([1, 0, 1, 0])
    a, p = Acc(), PerClassAcc()
    assert a.accumulate(out, tgt) == 0.75               # tie at row 3 -> first index, as torch.argmax
    assert p.accumulate(out, tgt) == {1: 0.5, 0: 1.0}
    res = collect_metrics({"Acc": a, "PerClassAcc": p}, {0: "zero", 1: "one", 2: "two"})
    assert res == {"metric_Acc": 0.75, "metric_PerClassAcc": {"one": 0.5, "zero": 1.0}}
    a.reset_metric(); p.reset_metric()
    assert a.total == 0 and p.total == {}


def test_checkpoint_loader_strips_dataparallel_prefix(tmp_path):
    from honk2_amd.utils import load_checkpoint_state
    path = tmp_path / "best_model.pt"
    torch.save({"model_state_dict": {"module.layers.output.bias": torch.ones(3)}, "best_epoch": 4}, path)
    sd, extra = load_checkpoint_state(str(path))
    assert list(sd) == ["layers.output.bias"] and extra["best_epoch"] == 4


def test_reference_written_checkpoint_loads(tmp_path):
    """tests/golden/checkpoint_res8_narrow_best_model.pt was written by the reference's own Workspace._save
    (oracle/gen_golden.py): DataParallel 'module.' prefixes, loss function and metric objects pickled by module
    reference (loss_function.ce_loss, metric.acc.Acc).  The reference's own loader fails on it under torch >= 2.6."""
    from honk2_amd.loss_function import ce_loss
    from honk2_amd.metric import Acc, PerClassAcc
    from honk2_amd.utils import load_checkpoint_state
    from oracle import weights
    sd, extra = load_checkpoint_state(os.path.join(GOLDEN, "checkpoint_res8_narrow_best_model.pt"))
    _, name, cfg, want, _, _ = load_golden_model("model_resnet__res8_narrow.npz")
    assert list(sd.keys()) == list(want.keys())
    for k, v in want.items():
        assert np.array_equal(sd[k].numpy(), v), k
    assert extra["best_epoch"] == 3 and extra["best_dev_loss"] == 1.25
    assert extra["loss_fn"] is ce_loss and isinstance(extra["metrics"]["Acc"], Acc)
    assert isinstance(extra["metrics"]["PerClassAcc"], PerClassAcc)
    model = find_cls(f"model.{name}")(dict(cfg))
    model.load_state_dict(sd, strict=True)


# ------------------------------------------------------------------ streaming dataset mirror
@pytest.mark.parametrize("tag,case", [("w1000_s10", dict(window_size_ms=1000, shift_size_ms=10, num_files=9, seed=77)),
                                      ("w400_s30", dict(window_size_ms=400, shift_size_ms=30, num_files=14, seed=5))])
def test_streaming_dataset_matches_the_reference_class(tag, case):
    """tests/golden/streaming_dataset.npz was produced by the reference's own StreamingDataset (dataset/dataset_utils.py:
    20-98) over the same synthetic utterances (oracle/gen_golden_streaming.py): same windows, same majority-label targets,
    same length -- read here out of order, which the reference cannot do."""
    import random
    from honk2_amd.dataset import SyntheticStreamingDataset
    z = np.load(os.path.join(GOLDEN, "streaming_dataset.npz"))
    cfg = dict(sample_rate=16000, target_class=["yes", "no", "up"], unknown_class=True, silence_class=True, type="dev", **case)
    random.seed(1234)
    ds = SyntheticStreamingDataset(cfg)
    n = int(z[f"{tag}_len"])
    assert len(ds) == n
    order = np.random.default_rng(0).permutation(n)[:200]
    for i in order:
        w, t = ds[int(i)]
        assert t == int(z[f"{tag}_targets"][i])
        assert len(w) == ds.window_size and w.dtype == np.float32
        assert abs(float(np.asarray(w, np.float64).sum()) - float(z[f"{tag}_sums"][i])) < 1e-9
        assert (float(w[0]), float(w[-1])) == tuple(z[f"{tag}_ends"][i])
    stream, window, shift, targets = ds.stream_view()
    assert (window, shift) == (ds.window_size, ds.shift_size)
    assert np.array_equal(targets, z[f"{tag}_targets"])
    assert len(stream) == (n - 1) * shift + window
    w5, _ = ds[5]
    assert np.array_equal(stream[5 * shift:5 * shift + window], w5)


def _replicate_like_torch(model):
    """What torch.nn.parallel.replicate() makes of `model` for ONE device (torch 2.10, nn/parallel/replicate.py), with clones
    standing in for the broadcast copies (no GPU here): every module is shallow-copied by _replicate_for_data_parallel() -- EMPTY
    _parameters --, children are re-linked, each parameter copy is set as a PLAIN ATTRIBUTE and recorded in _former_parameters,
    buffers are replaced by copies."""
    modules = list(model.modules())
    index = {m: i for i, m in enumerate(modules)}
    copies = []
    for m in modules:
        r = m._replicate_for_data_parallel()
        r._former_parameters = {}
        copies.append(r)
    for i, m in enumerate(modules):
        r = copies[i]
        for key, child in m._modules.items():
            r._modules[key] = None if child is None else copies[index[child]]
        for key, param in m._parameters.items():
            if param is None:
                r._parameters[key] = None
            else:
                c = param.detach().clone()
                setattr(r, key, c)
                r._former_parameters[key] = c
        for key, buf in m._buffers.items():
            r._buffers[key] = None if buf is None else buf.detach().clone()
    return copies[0]


@pytest.mark.parametrize("family", ["ResNet", "CNN"])
def test_data_parallel_replicas_get_one_engine_per_device_and_never_close_each_others(monkeypatch, family):
    """nn.DataParallel replicas (reference multi-GPU mechanism, run/test.py:69-70) have no parameters() and a state_dict() without
    the weights (see _replicate_like_torch): the replica on cuda:1 must still build its own engine, load it with ALL tensors of the
    module it was made from, key the upload on that module's tensor versions, and leave the cuda:0 engine alone."""
    from honk2_amd.model import model_utils

    class FakeEngine:
        created, closed = [], []

        def __init__(self, desc, device):
            self.device = torch.device(device)
            self.loaded = []
            FakeEngine.created.append(self)

        def load_tensor(self, name, tensor):
            self.loaded.append(name)

        def close(self):
            FakeEngine.closed.append(self)

    current = {"index": 0}
    monkeypatch.setattr(model_utils._lib, "Engine", FakeEngine)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: current["index"])
    if family == "ResNet":
        cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    else:
        cfg = {"time": 101, "frequency": 40, "dropout_prob": 0.5, "n_labels": 12,
               "conv_0": {"out_channels": 64, "kernel_size": [20, 8], "stride": [1, 1]}, "pool_0": {"kernel_size": [2, 2]},
               "conv_1": {"out_channels": 64, "kernel_size": [10, 4], "stride": [1, 1]}, "pool_1": {"kernel_size": [1, 1]}}
    model = find_cls(f"model.{family}")(dict(cfg)).eval()
    names = list(model.state_dict())
    replica = _replicate_like_torch(model)
    assert list(replica.parameters()) == [] and len(replica.state_dict()) < len(names)     # what broke engine() before
    e0 = model.engine()
    current["index"] = 1
    e1 = replica.engine()
    current["index"] = 0
    assert e0 is not e1 and (e0.device.index, e1.device.index) == (0, 1)
    assert e0.loaded == names and e1.loaded == names                      # the replica's engine got every tensor
    assert model.engine() is e0 and FakeEngine.closed == [] and len(FakeEngine.created) == 2
    assert model.engine() is e0 and e0.loaded == names                    # unchanged weights are not uploaded again
    # the next forward of DataParallel makes NEW replicas (fresh copies, possibly at recycled addresses, all at version 0): no re-upload ...
    current["index"] = 1
    replica2 = _replicate_like_torch(model)
    assert replica2.engine() is e1 and e1.loaded == names
    # ... until the SOURCE module's weights change: then every device's engine reloads, whatever the copies' addresses are
    with torch.no_grad():
        next(model.parameters()).add_(1.0)
    assert _replicate_like_torch(model).engine() is e1 and e1.loaded == 2 * names
    current["index"] = 0
    assert model.engine() is e0 and e0.loaded == 2 * names


def test_bench_parity_sample_is_spread_in_every_prefix():
    """bench.py's CPU leg evaluates a time-bounded PREFIX of its sample: every prefix has to straddle the whole batch."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    idx = bench.spread_indices(65536, 16384)
    assert len(np.unique(idx)) == 16384 and idx.min() == 0 and idx.max() == 65532
    for n in (256, 1024, 8192, 13056):
        pre = np.sort(idx[:n])
        assert pre[0] < 65536 // 8 and pre[-1] > 65536 - 65536 // 8          # reaches both ends
        assert np.diff(pre).max() <= 2 * 65536 // (1 << (n.bit_length() - 1))   # no hole wider than two steps of the largest even grid inside the prefix
    lo, hi = [], []
    from honk2_amd import dist_utils
    for r in range(8):
        a, b = dist_utils.shard_bounds(65536, r, 8)
        lo.append(a); hi.append(b)
    assert lo[0] == 0 and hi[-1] == 65536 and all(b - a == 8192 for a, b in zip(lo, hi))   # the shard record's 8 192 clips


def test_models_deepcopy_and_pickle_without_their_engines():
    """copy.deepcopy(model) / pickle (torch.save of a whole module) carry the tensors only; the copy builds engines of its own."""
    import copy
    import pickle
    cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
    model = find_cls("model.ResNet")(dict(cfg)).eval()
    model._engines[0] = ["sentinel", None]
    for clone in (copy.deepcopy(model), pickle.loads(pickle.dumps(model))):
        assert clone._engines == {} and clone._engines_lock is not model._engines_lock
        assert all(torch.equal(a, b) for a, b in zip(clone.state_dict().values(), model.state_dict().values()))
        assert not clone.training
    assert model._engines[0][0] == "sentinel"


def test_header_is_plain_c_and_the_c_client_links(tmp_path):
    """include/kws.h must compile as C99 with nothing but <stddef.h> / <stdint.h>, and tests/c_abi/kws_c_client.c must link against
    libkws_hip.so + the HIP runtime alone (the -m gpu suite runs it)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    probe = tmp_path / "probe.c"
    probe.write_text('#include "kws.h"\nint main(void) { kws_model_desc d; d.struct_size = (int32_t)sizeof d; return d.struct_size == 0 || KWS_ABI_VERSION == 0 || KWS_OK != 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(probe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = os.path.join(ROOT, "honk2_amd", "libkws_hip.so")
    if not (os.path.exists(lib) and os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h")):
        pytest.skip("library not built or no HIP headers")
    r = subprocess.run(["gcc", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "c_abi", "kws_c_client.c"), "-o", str(tmp_path / "kws_c_client"),
                        "-L" + os.path.join(ROOT, "honk2_amd"), "-lkws_hip", "-L/opt/rocm/lib", "-lamdhip64",
                        "-Wl,-rpath," + os.path.join(ROOT, "honk2_amd"), "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
