#!/usr/bin/env python3
"""Time the fused res8 kernel alone (features -> logits) on 65 536 clips with HIP events; optional env: KWS_LIB (another
build of the library), KWS_R8_DEBUG (ablations: 1 skip conv_0, 2 skip the k-loops), KWS_R8_WGS_PER_CU."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from honk2_amd.utils import find_cls

RES8 = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
B = int(os.environ.get("R8_B", "65536"))
torch.manual_seed(0)
model = find_cls("model.ResNet")(dict(RES8))
sd = model.state_dict()
for k, v in sd.items():
    if k.endswith("running_mean"): sd[k] = 0.3 + 0.2 * torch.randn_like(v)
    elif k.endswith("running_var"): sd[k] = 0.25 + 0.5 * torch.rand_like(v)
model.load_state_dict(sd)
model = model.cuda().eval()
x = torch.randn(B, 101, 40, device="cuda") * 2.5 + 0.65
import time
for _ in range(3): y = model(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < float(os.environ.get("R8_SETTLE_S", "0.3")):   # the clock governor settles over ~100 ms of load (tools/ramp_probe.sh)
    y = model(x)
    torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
reps = int(os.environ.get("R8_REPS", "10"))
ev[0].record()
for _ in range(reps): y = model(x)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
print(json.dumps({"tag": os.environ.get("R8_TAG", ""), "lib": os.path.basename(os.environ.get("KWS_LIB", "default")), "debug": os.environ.get("KWS_R8_DEBUG", "0"),
                  "wgs": os.environ.get("KWS_R8_WGS_PER_CU", "2"), "ms": round(ms, 3), "TFLOPs": round(74.35e6 * B / ms / 1e9, 1),
                  "frac_839": round(74.35e6 * B / ms / 1e9 / 838.7, 3), "checksum": float(y.double().abs().sum())}), flush=True)
