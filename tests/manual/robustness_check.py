"""Manual check (GPU box, from the repo root): many random weight / input draws and batch sizes through the fused res8
path against the oracle's torch engine."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import models, weights, frontend
from honk2_amd.utils import find_cls
def build(name, cfg, sd):
    m = find_cls(f"model.{name}")(dict(cfg)); m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}); return m.to("cuda:0").eval()
cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
worst = 0
for seed in range(5):
    sd = weights.make_state_dict("ResNet", cfg, seed=100 + seed)
    m = build("ResNet", cfg, sd)
    for B in (1, 2, 255, 256, 257, 511, 512, 513, 1031):
        feats = weights.make_features(B, seed=seed * 37 + B)
        if B > 3:
            feats[1] *= 12.0          # loud clip: features up to +-100
            feats[2] *= 1e-3          # nearly silent features
        y = m(torch.from_numpy(feats).cuda()).cpu().numpy()
        want = models.forward_torch("ResNet", cfg, sd, feats).numpy()
        err = np.abs(y - want).max() / max(1.0, np.abs(want).max())
        worst = max(worst, err)
        assert err < 1e-4 and np.isfinite(y).all(), (seed, B, err)
        top = np.sort(want, 1); clear = (top[:, -1] - top[:, -2]) > 1e-4
        assert (y.argmax(1) == want.argmax(1))[clear].all()
print("fused res8 robustness ok, worst rel err", worst)
# tiny BN variance -> large activations (fp16 range): var 1e-4 -> rstd 100
sd = weights.make_state_dict("ResNet", cfg, seed=5)
for k in sd:
    if k.endswith("running_var"): sd[k] = (sd[k] * 0 + 2e-4).astype(np.float32)
m = build("ResNet", cfg, sd); feats = weights.make_features(64, seed=9)
y = m(torch.from_numpy(feats).cuda()).cpu().numpy(); want = models.forward_torch("ResNet", cfg, sd, feats).numpy()
print("tiny-variance BN: max|logit|", np.abs(want).max(), "rel err", np.abs(y - want).max() / np.abs(want).max(), "finite", np.isfinite(y).all())
