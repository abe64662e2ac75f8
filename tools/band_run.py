#!/usr/bin/env python3
"""Runs cnn-trad-pool2 (DT = f32 | fp16) on 1 024 clips so that a -DBAND_TIMING build (KWS_LIB) can dump conv_band_kernel's phase stamps
(KWS_BAND_TIMING=<file>); read them with tools/band_phases.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from honk2_amd.utils import find_cls
cfg = {"time": 101, "frequency": 40, "dropout_prob": 0.5, "n_labels": 12, "dtype": os.environ.get("DT", "f32"),
       "conv_0": {"out_channels": 64, "kernel_size": [20, 8], "stride": [1, 1]}, "pool_0": {"kernel_size": [2, 2]},
       "conv_1": {"out_channels": 64, "kernel_size": [10, 4], "stride": [1, 1]}, "pool_1": {"kernel_size": [1, 1]}}
torch.manual_seed(3)
m = find_cls("model.CNN")(cfg).cuda().eval()
x = torch.randn(1024, 101, 40, device="cuda") * 2.5 + 0.65
for _ in range(2):
    y = m(x)
torch.cuda.synchronize()
