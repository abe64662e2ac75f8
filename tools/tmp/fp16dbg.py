import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["KWS_FORCE_LAYERWISE"] = "1"
import numpy as np, torch
from honk2_amd.utils import find_cls
def build(cfg, sd=None):
    torch.manual_seed(3)
    m = find_cls("model.ResNet")(dict(cfg))
    if sd is None:
        sd = m.state_dict()
        for k, v in sd.items():
            if k.endswith("running_mean"): sd[k] = 0.3 + 0.2 * torch.randn_like(v)
            elif k.endswith("running_var"): sd[k] = 0.25 + 0.5 * torch.rand_like(v)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd
x = torch.randn(64, 101, 40, device="cuda") * 2.5 + 0.65
for nl in (1, 2, 3, 4, 6):
    for dil in (False, True):
        cfg = {"n_feature_maps": 45, "n_layers": nl, "use_dilation": dil, "n_labels": 12}
        m32, sd = build(dict(cfg, dtype="f32"))
        want = m32(x).cpu().numpy()
        for dt in ("fp16", "bf16"):
            m, _ = build(dict(cfg, dtype=dt), sd)
            got = m(x).cpu().numpy()
            print(nl, dil, dt, m.plan_name(), "err", np.abs(got - want).max(), "max", np.abs(want).max(), flush=True)
