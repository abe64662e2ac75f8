"""Phase stamps (cycle counter) of the front-end workgroups: needs a -DFE16_TIMING build (KWS_LIB=honk2_amd/variants/lib_fets.so);
the stamps overwrite part of the feature rows.  KWS_FE_WGS_PER_CU=1 shows a lone workgroup's phases."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from honk2_amd.utils import AudioProcessor
ap = AudioProcessor()
B = int(os.environ.get("FE_B", "65536"))
wav = (0.1 * torch.randn(B, 16000, device='cuda')).clamp(-1, 1)
ap.compute_mfccs_batch(wav)
f = ap.compute_mfccs_batch(wav).cpu().numpy()
# stamps 0..7 + end (8) in program order, then the finer stamps 8..12 of the squares / power-tile phase (stored at 9..13)
order = [0, 1, 2, 3, 9, 10, 11, 12, 13, 4, 5, 7, 8]
names = ['top(A loads)', 'barrier', 'kloop', 'squares', 'barrier(image dead)', 'P store+barrier', 'P read-add-write+dma wait', 'barrier', 'issue next', 'mel', 'log', 'stage_next+store']
acc = np.zeros(len(names))
n = 0
for clip in range(1000, B, 997):
    for w in range(4):
        ts = f[clip].reshape(-1)[40 * (1 + w * 10): 40 * (1 + w * 10) + 28].view(np.uint64).astype(np.int64)
        t = ts[order]
        d = np.diff(t)
        if (d < 0).any() or d.sum() > 1e7:
            continue
        acc += d
        n += 1
print(os.environ.get("KWS_FE_WGS_PER_CU", "2"), "WG/CU;", n, "samples; mean cycles per phase:")
print(' | '.join(f'{nm}={x / n:.0f}' for nm, x in zip(names, acc)), '| total', f'{acc.sum() / n:.0f}')
