"""Three front-end launches at the bench batch, for rocprofv3 counter passes."""
import sys, torch
sys.path.insert(0, '.')
from honk2_amd.utils import AudioProcessor
ap = AudioProcessor()
wav = (0.1 * torch.randn(65536, 16000, device='cuda')).clamp(-1, 1)
for _ in range(3):
    f = ap.compute_mfccs_batch(wav)
torch.cuda.synchronize()
