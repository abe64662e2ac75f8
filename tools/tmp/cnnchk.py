import sys, os, glob, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import load_golden_model
from test_gpu_parity import _build
for f in sorted(glob.glob("/root/repo/tests/golden/model_cnn__*.npz")):
    tag, name, cfg, sd, feats, z = load_golden_model(os.path.basename(f))
    x = torch.from_numpy(feats).cuda()
    want = z["logits"]
    for dt in ("f32", "fp16"):
        m = _build(torch, name, dict(cfg, dtype=dt), sd)
        got = m(x).cpu().numpy()
        print(tag, dt, m.plan_name(), "err", float(np.abs(got - want).max()), "max|want|", float(np.abs(want).max()), flush=True)
