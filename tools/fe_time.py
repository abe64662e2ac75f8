#!/usr/bin/env python3
"""Time the front-end kernel alone (65 536 one-second clips, wav -> features) with HIP events; KWS_LIB selects another build."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from honk2_amd.utils import AudioProcessor
ap = AudioProcessor()
B = int(os.environ.get("FE_B", "65536"))
g = torch.Generator(device="cuda").manual_seed(3)
wav = (0.1 * torch.randn(B, 16000, device="cuda", generator=g)).clamp(-1, 1)
ap.compute_mfccs_batch(wav[:1024])
import time
for _ in range(3): f = ap.compute_mfccs_batch(wav)
torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < float(os.environ.get("FE_SETTLE_S", "0.3")):   # the clock governor settles over ~100 ms of load (tools/ramp_probe.sh)
    f = ap.compute_mfccs_batch(wav)
    torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
reps = int(os.environ.get("FE_REPS", "10"))
ev[0].record()
for _ in range(reps): f = ap.compute_mfccs_batch(wav)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
print(json.dumps({"tag": os.environ.get("FE_TAG", ""), "lib": os.path.basename(os.environ.get("KWS_LIB", "default")), "ms": round(ms, 3),
                  "GBps_alg": round(80160 * B / ms / 1e6, 1), "frac_8TBps": round(80160 * B / ms / 1e6 / 8000, 3),
                  "checksum": float(f.double().abs().sum())}), flush=True)
