#!/usr/bin/env python3
"""Experiment: consecutive steps alternate between two streams / two engines (own workspaces), so the tail of step i's res8 kernel
overlaps with the head of step i + 1's front end.  usage: two_stream_probe.py <batch> [serial|pipe]"""
import copy, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from honk2_amd.utils import find_cls
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mode = sys.argv[2] if len(sys.argv) > 2 else "pipe"
RES8 = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
torch.manual_seed(0)
m0 = find_cls("model.ResNet")(dict(RES8)).cuda().eval()
m1 = copy.deepcopy(m0)
wav = (0.1 * torch.randn(B, 16000, device="cuda")).clamp_(-1, 1)
outs = [torch.empty(B, 12, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
models = [m0, m1]
for m in models: m.forward_wav(wav, out=outs[0])
torch.cuda.synchronize()
steps = 40 if B <= 8192 else 14
def run(n):
    if mode == "serial":
        for i in range(n): m0.forward_wav(wav, out=outs[0])
    else:
        for i in range(n):
            with torch.cuda.stream(streams[i & 1]):
                models[i & 1].forward_wav(wav, out=outs[i & 1])
    torch.cuda.synchronize()
run(4)
t0 = time.perf_counter(); run(steps); dt = time.perf_counter() - t0
print(json.dumps({"batch": B, "mode": mode, "ms_per_step": round(1e3 * dt / steps, 4), "clips_per_s": round(B * steps / dt)}))
