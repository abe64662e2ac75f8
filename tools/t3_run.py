#!/usr/bin/env python3
"""Runs res15 (DT = f32 | bf16 | fp16) on 1 024 clips so that a -DT3_TIMING build (KWS_LIB) can dump the tiled 3x3 kernel's phase stamps
(KWS_T3_TIMING=<file>, KWS_T3_TIMING_LAYER=<i>); read them with tools/t3_phases.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from honk2_amd.utils import find_cls
cfg = {"n_feature_maps": 45, "n_layers": 13, "use_dilation": True, "n_labels": 12, "dtype": os.environ.get("DT", "f32")}
torch.manual_seed(3)
m = find_cls("model.ResNet")(cfg).cuda().eval()
x = torch.randn(1024, 101, 40, device="cuda") * 2.5 + 0.65
for _ in range(2):
    y = m(x)
torch.cuda.synchronize()
