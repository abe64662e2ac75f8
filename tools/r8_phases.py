"""Phase timestamps (s_memtime) of the fused res8 kernel's workgroups.  Needs a library whose res8_f16x3.hip was built
with -DR8H_TIMING: the kernel then parks eight timestamps per wave in the (consumed) feature rows of each clip."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from conftest import load_golden_model
from honk2_amd.utils import find_cls
tag, name, cfg, sd, feats, z = load_golden_model("model_resnet__res8.npz")
model = find_cls(f"model.{name}")(dict(cfg))
model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
model = model.to("cuda:0").eval()
B = 65536
x = (0.65 + 2.5 * torch.randn(B, 101, 40, device='cuda')).contiguous()
model(x); torch.cuda.synchronize()
x2 = (0.65 + 2.5 * torch.randn(B, 101, 40, device='cuda')).contiguous()
model(x2); torch.cuda.synchronize()
f = x2.cpu().numpy().reshape(B, -1)
names = ['stage', 'conv0', 'guard+map', 'layer0', 'layer1', 'layers2-4', 'layer5+tail']
for clip in (5, 20000, 40000, 65000):
    for w in range(4):
        ts = f[clip, 16 * w: 16 * w + 16].view(np.uint64).astype(np.int64)
        d = np.diff(ts)
        print(clip, w, ' '.join(f'{n}={int(v)}' for n, v in zip(names, d)), 'total', int(ts[-1] - ts[0]))
