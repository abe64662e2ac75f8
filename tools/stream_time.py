#!/usr/bin/env python3
"""Streaming front end / wav->logits over the sliding windows of one long stream: shared-frame path vs every window on its own
(KWS_WINDOWS_NO_SHARE=1).  1000 ms windows, 10 ms shift (config/gsc_dev_config.json:62-63 of the reference)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from honk2_amd.utils import AudioProcessor, find_cls

n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
window, shift = 16000, 160
stream = (0.1 * torch.randn((n_win - 1) * shift + window, device="cuda")).clamp(-1, 1)
ap = AudioProcessor()
cfg = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
torch.manual_seed(7)
model = find_cls("model.ResNet")(dict(cfg))          # default init: timing only
model = model.to("cuda:0").eval()
for name, fn in (("mfcc_windows", lambda: ap.compute_mfccs_windows(stream, window, shift)),
                 ("forward_windows", lambda: model.forward_windows(stream, window, shift))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt * 1e3:.2f} ms for {n_win} windows = {n_win / dt / 1e6:.2f} M windows/s", flush=True)
