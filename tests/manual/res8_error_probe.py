"""Manual check (GPU box, from the repo root): the three fused res8 kernels against a float64 evaluation of the oracle."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden_model
from oracle import models
from honk2_amd.utils import find_cls
tag, name, cfg, sd, feats, z = load_golden_model("model_resnet__res8.npz")
exact = models.forward_numpy(name, cfg, sd, feats, np.float64)
print("fp32 torch CPU vs f64:", np.abs(models.forward_torch(name, cfg, sd, feats).numpy() - exact).max())
for impl in ("", "bf16x6", "fp32"):
    if impl: os.environ["KWS_RES8_IMPL"] = impl
    m = find_cls(f"model.{name}")(dict(cfg)); m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}); m = m.to("cuda:0").eval()
    y = m(torch.from_numpy(feats).cuda()).cpu().numpy()
    print(m.plan_name(), "vs f64: max", np.abs(y - exact).max(), "rms", np.sqrt(np.mean((y - exact) ** 2)))
