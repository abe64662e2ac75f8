#!/usr/bin/env python3
"""Headline benchmark: 1-s clips/sec end-to-end (wav -> logits), res8, GSCv2 shapes, on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (kws_forward_wav: MFCC front end + res8) over the GLOBAL batch of synthetic
16 kHz one-second clips (default 65 536, BASELINE.json configs[3] / the north-star target batch), sharded
contiguously over the N ranks (strong scaling: 65 536 / N clips per GPU), followed for N > 1 by the RCCL
all-gather of the (B/N, 12) logits.  Waveforms are resident in HBM before the timed region starts.  Rank 0 prints
ONE JSON line.  `roofline` is for the dominant kernel (the fused res8 kernel, fp32 matrix cores): algorithmic
FLOPs per launch / its mean launch duration measured with HIP events on the launch stream inside the timed steps.
`cpu_baseline` (rank 0, N = 1 only) times the CPU oracle ("port": numpy/scipy rFFT front end + torch-CPU fp32
model, parity-pinned against the reference in tests/) on a bounded sample of the same clips.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RES8 = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
F_ALG_MODEL = 74.35e6          # FLOP / clip, conv + linear, 2 x MAC
F_ALG_FRONTEND = 1.22e6        # FLOP / clip, FFT-based count
B_ALG = 64048                  # HBM bytes / clip end to end: 16 000 fp32 samples in + 12 fp32 logits out
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32-input matrix (= fp32 vector) peak
PEAK_BF16_MFMA_TFLOPS = 2516.0 # MI355X_MICROARCH.md: dense bf16 matrix peak (spec); 2.0 PF sustained on this box (tools/coexec_probe_bf16)
PEAK_HBM_GBS = 8000.0


def synth_wav(torch, n, seed, device):
    """SURVEY.md section 8d inputs, generated on the device: 0.1*randn clamped to [-1,1]; every 12th clip exact
    zeros; every 12th+1 a 0.5-amplitude 1 kHz tone (carrying the clip's noise at -37 dB as dither)."""
    g = torch.Generator(device=device).manual_seed(seed)
    wav = torch.empty((n, 16000), dtype=torch.float32, device=device)
    for lo in range(0, n, 8192):
        hi = min(n, lo + 8192)
        wav[lo:hi] = (0.1 * torch.randn((hi - lo, 16000), generator=g, device=device)).clamp_(-1, 1)
    t = torch.arange(16000, device=device, dtype=torch.float64) / 16000.0
    tone = (0.5 * torch.sin(2 * torch.pi * 1000.0 * t)).float()
    wav[1::12] = tone + 0.05 * wav[1::12]
    wav[0::12] = 0
    return wav


def build_model(torch, device):
    import numpy as np
    from honk2_amd.utils import find_cls
    from oracle import weights   # deterministic weights shared with the cpu_baseline leg (bench-only use of oracle/)
    sd = weights.make_state_dict("ResNet", RES8, seed=0)
    model = find_cls("model.ResNet")(dict(RES8))
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    return model.to(device).eval(), sd


def cpu_baseline(torch, wav_sample, sd, budget_s=12.0):
    """Time the CPU oracle on host cores over a bounded sample (about 10-20 s of CPU work)."""
    import numpy as np
    from oracle import frontend, models
    # the one-GPU box exposes every host core but grants a 16-worker share; stay inside the affinity mask and that share
    threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(threads)
    x = wav_sample.cpu().numpy()

    def run(chunk):
        feats = frontend.compute_mfccs_batch(chunk, "f32")
        return models.forward_torch("ResNet", RES8, sd, feats)

    run(x[:64])                                     # warm-up (thread pools, allocator)
    t0 = time.perf_counter()
    run(x[:256])
    per_clip = (time.perf_counter() - t0) / 256
    n = int(min(len(x), max(256, budget_s / max(per_clip, 1e-9))))
    n = max(256, n // 256 * 256)
    t0 = time.perf_counter()
    for lo in range(0, n, 1024):
        run(x[lo:lo + 1024])
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"{n} of the benchmark's clips, chunks of 1024, numpy/scipy rFFT front end (fp32) + torch-CPU fp32 res8, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="GLOBAL batch (clips per step over all GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from honk2_amd import dist_utils

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one process per GPU)")
    # KWS_BENCH_BACKEND / KWS_BENCH_ONE_DEVICE are rehearsal knobs only (several ranks on one GPU over gloo, to exercise
    # the N > 1 code path on a single-GPU box); the real multi-GPU run uses RCCL with one GPU per rank.
    rank, world = dist_utils.init_from_env(os.environ.get("KWS_BENCH_BACKEND", "nccl"))
    local = 0 if os.environ.get("KWS_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    lo, hi = dist_utils.shard_bounds(args.batch, rank, world)
    nloc = hi - lo
    model, sd = build_model(torch, device)
    wav = synth_wav(torch, nloc, 1234 + rank, device)
    logits = torch.empty((nloc, RES8["n_labels"]), dtype=torch.float32, device=device)
    counts = [b - a for a, b in (dist_utils.shard_bounds(args.batch, r, world) for r in range(world))]
    gathered = torch.empty((args.batch, RES8["n_labels"]), dtype=torch.float32, device=device) if world > 1 else None

    def step():
        model.forward_wav(wav, out=logits)
        if world > 1:
            if len(set(counts)) == 1:
                dist.all_gather_into_tensor(gathered, logits)     # RCCL over xGMI: the path's only collective
            else:
                gathered.copy_(dist_utils.all_gather_rows(logits, counts))

    for _ in range(args.warmup):
        step()
    engine = model.engine()
    torch.cuda.synchronize()
    engine.profile_enable(True)
    engine.profile_read()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    model_ms, front_ms, calls = engine.profile_read()
    engine.profile_enable(False)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    assert torch.isfinite(logits).all()

    if rank == 0:
        clips_per_s = args.batch * args.steps / elapsed
        k_ms = model_ms / max(calls, 1)
        f_ms = front_ms / max(calls, 1)
        achieved = F_ALG_MODEL * nloc / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        plan = model.plan_name()
        if plan == "res8_fused":
            # conv_1..6 run on the fp16 matrix cores with fp32-accurate 3-term products (x = x1 + x2 in fp16, weights
            # pre-scaled by a power of two; a2 b1 + a1 b2 + a1 b1, fp32 accumulate; more accurate than an fp32 FMA chain):
            # every algorithmic MAC costs three fp16 MACs, so the roof for this arithmetic is the dense fp16 MFMA peak / 3.
            peak, kern = PEAK_BF16_MFMA_TFLOPS / 3.0, "res8h_kernel (fused conv stack, fp16 MFMA x 3 terms, fp32 accumulate)"
            note = ("peak = 2516 TFLOP/s dense fp16 / 3 terms; achieved is %.2fx the fp32-input MFMA roof of 157.3 TFLOP/s"
                    % (achieved / PEAK_F32_MFMA_TFLOPS))
        elif plan == "res8_fused_bf16x6":
            peak, kern = PEAK_BF16_MFMA_TFLOPS / 6.0, "res8x_kernel (fused conv stack, bf16 MFMA x 6 terms, fp32 accumulate)"
            note = ("peak = 2516 TFLOP/s dense bf16 / 6 terms; achieved is %.2fx the fp32-input MFMA roof of 157.3 TFLOP/s"
                    % (achieved / PEAK_F32_MFMA_TFLOPS))
        else:
            peak, kern, note = PEAK_F32_MFMA_TFLOPS, "res8_kernel (fused conv stack, fp32-input MFMA)", ""
        roofline = {"bound": "mfma", "kernel": kern, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                    "frac": achieved / peak, "traffic": None, "kernel_ms": k_ms, "launches": calls,
                    "flop_per_launch": F_ALG_MODEL * nloc, "note": note}
        # HBM bytes per launch cannot be measured from inside the process; when the committed counter summary of this very
        # workload (same kernel, same clips per launch) is present, quote it: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate passes, KB units, x2 correction on the gfx950 fetch counter (MI355X_MICROARCH.md).
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "v8_summary.json")
        if plan == "res8_fused" and nloc == 65536 and os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    summ = json.load(f)
                ent = next(v for k, v in summ.items() if "res8h_kernel" in k)
                roofline["traffic"] = (2.0 * ent["FETCH_SIZE"] + ent["WRITE_SIZE"]) * 1024.0
                roofline["traffic_source"] = "profiles/r01/v8_summary.json (rocprofv3 --pmc, same command)"
            except Exception:
                pass
        out = {
            "metric": "1s-clips/sec end-to-end (wav->logits), res8 GSCv2", "value": clips_per_s, "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": ("f32 (f16x3: fp32-accurate 3-term fp16 products, fp32 accumulate, in the conv stack and in the front end's DFT)" if plan == "res8_fused" else "f32 (bf16x6: fp32-accurate 6-term bf16 products, fp32 accumulate; fp32 front end)" if plan == "res8_fused_bf16x6" else "f32"), "data": "synthetic",
            "config": {"workload": f"res8 fp32 wav->logits, global batch {args.batch} one-second 16 kHz clips "
                                   f"(BASELINE configs[3]), {nloc} clips/GPU, random-init weights",
                       "global_batch": args.batch, "clips_per_gpu": nloc, "n_samples": 16000,
                       "plan": model.plan_name(), "parallelism": f"dp{world} (clip sharding, logits all-gather)"},
            "roofline": roofline,
            "frontend": {"kernel": "frontend_kernel (STFT+mel+log, fp32 MFMA)", "kernel_ms": f_ms,
                         "hbm_GBps_algorithmic": (80160 * nloc / (f_ms * 1e-3) / 1e9) if f_ms > 0 else 0.0},
            "end_to_end": {"hbm_frac_of_8TBps": clips_per_s / world * B_ALG / (PEAK_HBM_GBS * 1e9),
                           "flop_frac_of_fp32_peak": clips_per_s / world * (F_ALG_MODEL + F_ALG_FRONTEND) / (PEAK_F32_MFMA_TFLOPS * 1e12)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch, wav[:16384], sd)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
