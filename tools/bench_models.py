#!/usr/bin/env python3
"""Throughput of every shipped model config through the C ABI (features -> logits and wav -> logits) on one GPU.
Not the headline benchmark (that is bench.py); used to fill the per-model table in DESIGN.md."""
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from honk2_amd.utils import find_cls

MFLOP = {"resnet__res8": 74.35, "resnet__res8_narrow": 14.05, "resnet__res15": 1917.63, "resnet__res15_narrow": 342.66,
         "resnet__res26": 878.07, "resnet__res26_narrow": 157.33, "cnn__cnn-trad-pool2": 192.37,
         "cnn__cnn-trad-fpool3": 249.19, "cnn__cnn-one-fpool3": 66.57, "cnn__cnn-one-fstride4": 62.54,
         "cnn__cnn-one-fstride8": 61.76, "cnn__cnn-tstride2": 152.43, "cnn__cnn-tstride4": 64.28,
         "cnn__cnn-tstride8": 34.05, "cnn__cnn-tpool2": 204.91, "cnn__cnn-tpool3": 147.40}


def main():
    dtype = os.environ.get("KWS_BENCH_DTYPE", "f32")      # f32 (fp32-accurate, three-term fp16 products) | bf16x3 | bf16 | fp16
    peak = {"f32": 2516.0 / 3, "bf16x3": 2516.0 / 3, "bf16": 2516.0, "fp16": 2516.0}[dtype]   # TFLOP/s roof of the dtype's arithmetic
    only = sys.argv[1:] or None
    dev = torch.device("cuda:0")
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "model_*.npz"))):
        tag = os.path.basename(path)[6:-4]
        if tag.startswith("hey_snips") or (only and tag not in only):
            continue
        z = np.load(path)
        name, cfg = str(z["model_name"]), json.loads(str(z["model_config"]))
        torch.manual_seed(7)                                   # the model's own default init; BN statistics made non-trivial
        model = find_cls(f"model.{name}")(dict(cfg, dtype=dtype))
        sd = model.state_dict()
        for k, v in sd.items():
            if k.endswith("running_mean"):
                sd[k] = 0.3 + 0.2 * torch.randn_like(v)
            elif k.endswith("running_var"):
                sd[k] = 0.25 + 0.5 * torch.rand_like(v)
        model.load_state_dict(sd)
        model = model.to(dev).eval()
        batch = int(os.environ.get("KWS_BENCH_BATCH", "0")) or (8192 if MFLOP[tag] < 400 else 2048)
        x = torch.randn(batch, 101, 40, device=dev) * 2.5 + 0.65
        if os.environ.get("KWS_BENCH_ZERO") == "1":       # all-zero features: the same instruction stream with (almost) no switching activity in the matrix pipe
            x.zero_()
        model(x[:64])
        model(x)                       # full-batch warm-up: the workspace grows to its final size here, not in the timed calls
        torch.cuda.synchronize()
        t0 = time.perf_counter()           # ~300 ms of untimed load (the clock governor settles over ~100 ms: tools/ramp_probe.sh), then ~250 ms timed
        model(x)
        torch.cuda.synchronize()
        one = max(time.perf_counter() - t0, 1e-4)
        for _ in range(min(300, int(0.3 / one) + 1)):
            model(x)
        torch.cuda.synchronize()
        reps = max(3, min(200, int(0.25 / one)))
        power = None
        if os.environ.get("KWS_BENCH_POWER") == "1":      # ~3 s of calls with rocm-smi sampled beside them: package power and shader clock under this model
            import re, subprocess, threading
            samples, stop = [], [False]

            def sample():
                while not stop[0]:
                    try:
                        samples.append(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout[:600])
                    except Exception as e:
                        samples.append(repr(e))
                    time.sleep(0.25)
            th = threading.Thread(target=sample)
            th.start()
            reps = max(reps, int(3.0 / one))
        t0 = time.perf_counter()
        for _ in range(reps):
            model(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        if os.environ.get("KWS_BENCH_POWER") == "1":
            stop[0] = True
            th.join()
            pw = [float(m.group(1)) for smp in samples[2:] for m in [re.search(r'Package Power \(W\)": "([0-9.]+)"', smp)] if m]
            ck = [float(m.group(1)) for smp in samples[2:] for m in [re.search(r'sclk clock speed:": "\(([0-9.]+)Mhz', smp)] if m]
            power = {"W": round(sum(pw) / max(len(pw), 1)), "sclk_MHz": round(sum(ck) / max(len(ck), 1)), "samples": len(pw)}
        cps = batch / dt
        print(json.dumps({"model": tag, "plan": model.plan_name(), "dtype": dtype, "batch": batch, "ms": round(dt * 1e3, 2),
                          "clips_per_s": round(cps), "TFLOPs_alg": round(cps * MFLOP[tag] * 1e6 / 1e12, 2),
                          "frac_of_roof": round(cps * MFLOP[tag] * 1e6 / 1e12 / peak, 3), "roof_TFLOPs": round(peak, 1), **({"power": power} if power else {})}), flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
