import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import load_golden_model
from test_gpu_parity import _build
for fname in ["model_resnet__res15.npz", "model_resnet__res26_narrow.npz", "model_resnet__res26.npz"]:
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    x = torch.from_numpy(feats).cuda()
    want = z["logits"]
    for dt in ("fp16", "bf16", "bf16x3", "f32"):
        got = _build(torch, name, dict(cfg, dtype=dt), sd)(x).cpu().numpy()
        print(tag, dt, "err", np.abs(got - want).max(), "max|want|", np.abs(want).max(), flush=True)
