import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import load_golden_model
from test_gpu_parity import _build
for fname in ["model_hey_snips__res26.npz", "model_resnet__res8_narrow.npz", "model_resnet__res15_narrow.npz", "model_resnet__res26.npz"]:
    tag, name, cfg, sd, feats, z = load_golden_model(fname)
    x = torch.from_numpy(feats).cuda()
    want = z["logits"]
    for dt in ("fp16", "bf16", "bf16x3"):
        m = _build(torch, name, dict(cfg, dtype=dt), sd)
        got = m(x).cpu().numpy()
        top = np.sort(want, axis=1)
        print(tag, dt, m.plan_name(), "err", float(np.abs(got - want).max()), "max|want|", float(np.abs(want).max()), "argmax_eq", bool((got.argmax(1) == want.argmax(1)).all()), flush=True)
