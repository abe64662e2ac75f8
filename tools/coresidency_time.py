"""Does a front-end workgroup next to a res8 workgroup on every CU beat two of a kind?  Runs the two kernels of the
wav -> logits path on two streams with ONE workgroup per CU each (KWS_FE_WGS_PER_CU=1 KWS_R8_WGS_PER_CU=1: both fit a CU
together, 79.6 + 78 KB of LDS) and compares with the default (each kernel alone, two workgroups per CU)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden_model
from honk2_amd.utils import find_cls, AudioProcessor
tag, name, cfg, sd, feats, z = load_golden_model("model_resnet__res8.npz")
model = find_cls(f"model.{name}")(dict(cfg))
model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
model = model.to("cuda:0").eval()
ap = AudioProcessor()
B = 65536
wav = (0.1 * torch.randn(B, 16000, device='cuda')).clamp(-1, 1)
feat = ap.compute_mfccs_batch(wav)
model(feat); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(concurrent, reps=4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        if concurrent:
            with torch.cuda.stream(s1): ap.compute_mfccs_batch(wav)
            with torch.cuda.stream(s2): model(feat)
        else:
            ap.compute_mfccs_batch(wav); model(feat)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print(os.environ.get('KWS_FE_WGS_PER_CU', '2'), os.environ.get('KWS_R8_WGS_PER_CU', '2'),
      'sequential ms', round(run(False), 2), 'two streams ms', round(run(True), 2))
