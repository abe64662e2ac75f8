import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_model_files():
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith("model_") and f.endswith(".npz"))


def load_golden_model(fname):
    """-> (tag, model_name, cfg, state_dict (numpy), feats, golden npz) with weights/inputs regenerated from the seed."""
    from oracle import weights
    z = np.load(os.path.join(GOLDEN, fname))
    name = str(z["model_name"])
    cfg = json.loads(str(z["model_config"]))
    sd = weights.make_state_dict(name, cfg, seed=int(z["seed"]))
    feats = weights.make_features(int(z["batch"]), seed=int(z["seed"]) + 1, time=int(z["time"]))
    return fname[6:-4], name, cfg, sd, feats, z


@pytest.fixture(scope="session")
def lib_built():
    from honk2_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


@pytest.fixture(scope="session")
def lib_experiments():
    """The EXPERIMENTS=1 build (debug / timing switches, fault injection: honk2_amd/csrc/Makefile), bound beside the product library."""
    import subprocess
    from honk2_amd import _lib
    subprocess.run(["make", "-C", os.path.join(ROOT, "honk2_amd", "csrc"), "EXPERIMENTS=1", "-j", str(min(8, os.cpu_count() or 1))],
                   check=True, capture_output=True)
    return _lib.bind(_lib.EXP_LIB_PATH)
