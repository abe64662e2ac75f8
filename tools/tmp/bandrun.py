import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import load_golden_model
from test_gpu_parity import _build
tag, name, cfg, sd, feats, z = load_golden_model("model_cnn__cnn-trad-pool2.npz")
m = _build(torch, name, dict(cfg, dtype=os.environ.get("DT", "f32")), sd)
x = torch.randn(1024, 101, 40, device="cuda") * 2.5 + 0.65
for _ in range(2): y = m(x)
torch.cuda.synchronize()
