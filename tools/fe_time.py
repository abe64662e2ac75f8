import sys, time, torch
sys.path.insert(0, '.')
from honk2_amd.utils import AudioProcessor
ap = AudioProcessor()
wav = (0.1 * torch.randn(65536, 16000, device='cuda')).clamp(-1, 1)
ap.compute_mfccs_batch(wav[:1024]); torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter()
    for _ in range(5):
        f = ap.compute_mfccs_batch(wav)
    torch.cuda.synchronize()
    print('FE ms', (time.perf_counter() - t0) / 5 * 1e3)
