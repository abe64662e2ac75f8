#!/usr/bin/env python3
"""conv3x3_stream.hip against the tile kernels on small models: which shape of run differs, and where."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from honk2_amd.utils import find_cls
from oracle import weights

def build(cfg, sd):
    m = find_cls("model.ResNet")(dict(cfg))
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    return m.to("cuda:0").eval()

dtype = os.environ.get("DT", "bf16")
if sys.argv[1:] == ["big"]:      # one res15 pass at 1 024 clips (tools/stream.sh timing: the STREAM_TIMING build prints its per-wave cycle counts)
    cfg = {"n_feature_maps": 45, "n_layers": 13, "use_dilation": True, "n_labels": 12}
    m = build(dict(cfg, dtype=dtype), weights.make_state_dict("ResNet", cfg, seed=11))
    m(torch.randn(1024, 101, 40, device="cuda")); torch.cuda.synchronize()
    sys.exit(0)
for nl, dil, n, t in ((1, False, 2, 101), (2, False, 2, 101), (3, False, 2, 101), (3, False, 300, 101), (6, True, 2, 101), (6, True, 300, 101), (1, False, 300, 101)):
    cfg = {"n_feature_maps": 45, "n_layers": nl, "use_dilation": dil, "n_labels": 12, "dtype": dtype}
    sd = weights.make_state_dict("ResNet", {k: v for k, v in cfg.items() if k != "dtype"}, seed=11)
    x = torch.from_numpy(weights.make_features(n, seed=12, time=t)).cuda()
    os.environ["KWS_T3_STREAM"] = "2"
    a = build(cfg, sd); ya = a(x); pa = a.plan_detail()
    os.environ["KWS_T3_STREAM"] = "0"; os.environ["KWS_T3_PAIR"] = "0"
    b = build(cfg, sd); yb = b(x)
    d = (ya - yb).abs()
    bad = (d.max(1).values > 0).nonzero().flatten().tolist()
    print(nl, dil, n, pa, "max diff", float(d.max()), "bad clips", len(bad), bad[:10], flush=True)
