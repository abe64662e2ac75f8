#!/usr/bin/env python3
"""Experiment: front end of batch i + 1 on a few CUs (CU-masked stream) while the fused res8 kernel of batch i runs on the rest.
usage: cu_split_probe.py <CUs for the front end> ; 0 = plain serial reference on the default stream"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fe_cus = int(sys.argv[1]) if len(sys.argv) > 1 else 48
import torch
torch.cuda.init(); torch.cuda.set_device(0)
from honk2_amd.utils import find_cls, AudioProcessor
hip = C.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count
B = 65536

def masked_stream(lo, hi):
    words = (NCU + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(lo, hi): mask[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

RES8 = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
if fe_cus:
    os.environ["KWS_N_CU"] = str(NCU - fe_cus)
torch.manual_seed(0)
model = find_cls("model.ResNet")(dict(RES8)).cuda().eval()
wav = (0.1 * torch.randn(B, 16000, device="cuda")).clamp_(-1, 1)
model(torch.zeros(4, 101, 40, device="cuda"))             # engine created with KWS_N_CU in force
if fe_cus: os.environ["KWS_N_CU"] = str(fe_cus)
ap = AudioProcessor()
f0 = ap.compute_mfccs_batch(wav[:64])                       # front-end engine created
os.environ.pop("KWS_N_CU", None)
torch.cuda.synchronize()
steps = 12
if not fe_cus:
    for _ in range(3): y = model.forward_wav(wav)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(steps): y = model.forward_wav(wav)
    ev[1].record(); torch.cuda.synchronize()
    print(json.dumps({"fe_cus": 0, "ms_per_step": ev[0].elapsed_time(ev[1]) / steps}))
    sys.exit(0)
sa, sb = masked_stream(0, NCU - fe_cus), masked_stream(NCU - fe_cus, NCU)
feats = [None, None]; fe_done = [None, None]; r8_done = [None, None]; outs = []
def run(n):
    for i in range(n + 1):
        b = i & 1
        if i < n:
            with torch.cuda.stream(sb):
                if r8_done[b] is not None: sb.wait_event(r8_done[b])
                feats[b] = ap.compute_mfccs_batch(wav)
                fe_done[b] = torch.cuda.Event(); fe_done[b].record(sb)
        if i >= 1:
            pb = (i - 1) & 1
            with torch.cuda.stream(sa):
                sa.wait_event(fe_done[pb])
                outs.append(model(feats[pb]))
                r8_done[pb] = torch.cuda.Event(); r8_done[pb].record(sa)
    torch.cuda.synchronize()
run(3); outs.clear()
import time
t0 = time.perf_counter(); run(steps); dt = time.perf_counter() - t0
print(json.dumps({"fe_cus": fe_cus, "ms_per_step": 1e3 * dt / steps}))
