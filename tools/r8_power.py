#!/usr/bin/env python3
"""Is the fused res8 kernel clock / power limited?  Times it on the usual random inputs and on all-zero operands (same
instruction stream, no switching activity in the matrix pipe) and samples rocm-smi while it loops.
env: R8_ZERO=1 -> zero features and zero conv weights."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from honk2_amd.utils import find_cls

RES8 = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
B = 65536
zero = os.environ.get("R8_ZERO", "0") == "1"
torch.manual_seed(0)
model = find_cls("model.ResNet")(dict(RES8))
sd = model.state_dict()
for k, v in sd.items():
    if k.endswith("running_mean"): sd[k] = 0.3 + 0.2 * torch.randn_like(v)
    elif k.endswith("running_var"): sd[k] = 0.25 + 0.5 * torch.rand_like(v)
    elif zero and "conv" in k: sd[k] = torch.zeros_like(v)
model.load_state_dict(sd)
model = model.cuda().eval()
x = torch.zeros(B, 101, 40, device="cuda") if zero else torch.randn(B, 101, 40, device="cuda") * 2.5 + 0.65
for _ in range(3): y = model(x)
torch.cuda.synchronize()
samples = []
stop = False
def sample():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out.strip()[:600])
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.3)
th = threading.Thread(target=sample); th.start()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
reps = int(os.environ.get("R8_REPS", "250"))
ev[0].record()
for _ in range(reps): y = model(x)
ev[1].record(); torch.cuda.synchronize()
stop = True; th.join()
ms = ev[0].elapsed_time(ev[1]) / reps
import re
pw = [float(m.group(1)) for smp in samples[2:] for m in [re.search(r'Package Power \(W\)": "([0-9.]+)"', smp)] if m]
ck = [float(m.group(1)) for smp in samples[2:] for m in [re.search(r'sclk clock speed:": "\(([0-9.]+)Mhz', smp)] if m]
pw_m = sum(pw) / max(len(pw), 1); ck_m = sum(ck) / max(len(ck), 1)
print(json.dumps({"zero": zero, "debug": os.environ.get("KWS_R8_DEBUG", "0"), "lib": os.path.basename(os.environ.get("KWS_LIB", "default")),
                  "ms": round(ms, 3), "power_W": round(pw_m, 1), "sclk_MHz": round(ck_m), "uJ_per_clip": round(pw_m * ms * 1e-3 / B * 1e6, 1),
                  "samples": len(pw)}))
