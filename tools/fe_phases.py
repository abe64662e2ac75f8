"""Phase timestamps (s_memtime) of the front-end workgroups: KWS_FE_DEBUG=8 build writes them into the feature rows."""
import os, sys, numpy as np, torch
os.environ['KWS_FE_DEBUG'] = '8'
sys.path.insert(0, '.')
from honk2_amd.utils import AudioProcessor
ap = AudioProcessor()
wav = (0.1 * torch.randn(65536, 16000, device='cuda')).clamp(-1, 1)
ap.compute_mfccs_batch(wav)
f = ap.compute_mfccs_batch(wav).cpu().numpy()
names = ['top', 'barrier', 'kloop', 'power+issue', 'mel', 'log', '-', 'stage_next+store']
for clip in (5, 20000, 40000, 65000):
    for w in range(4):
        row = f[clip, 1 + w * 8: 1 + w * 8 + 1 + 1].reshape(-1)[:16 * 1]
        ts = f[clip].reshape(-1)[40 * (1 + w * 10): 40 * (1 + w * 10) + 18].view(np.uint64)
        d = np.diff(ts.astype(np.int64))
        print(clip, w, ' '.join(f'{n}={int(x)}' for n, x in zip(names, d)), 'total', int(ts[-1] - ts[0]))
