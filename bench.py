#!/usr/bin/env python3
"""Headline benchmark: 1-s clips/sec end-to-end (wav -> logits), res8, GSCv2 shapes, on N MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: this process starts the N ranks itself, as fresh children)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                (the same ranks, started by the caller)

A step = one pass of the hot path (kws_forward_wav: MFCC front end + res8) over the GLOBAL batch of synthetic
16 kHz one-second clips (default 65 536, BASELINE.json configs[3] / the north-star target batch), sharded
contiguously over the N ranks (strong scaling: 65 536 / N clips per GPU), followed for N > 1 by the RCCL
all-gather of the (B/N, 12) logits.  Waveforms are resident in HBM before the timed region starts.  Rank 0 prints
ONE JSON line.  `roofline` is for the dominant kernel (the fused res8 kernel, fp32-accurate three-term fp16 products on the
matrix cores): algorithmic FLOPs per launch / its mean launch duration measured with HIP events on the launch stream
inside the timed steps.  `cpu_baseline` (rank 0, N = 1 only) times the CPU oracle ("port": numpy/scipy rFFT front end +
torch-CPU fp32 model, parity-pinned against the reference in tests/) on a bounded sample of the same clips, and `parity`
compares the GPU logits of exactly those clips with the oracle's (tolerance 1e-3, argmax); the run exits non-zero
when that check fails.  Inputs: SURVEY.md section 8d, except that the 1 kHz tone clips carry their clip's noise at
-37 dB as dither (a bare bin-centred sine leaves every off-peak mel band pure rounding noise, on which no two fp32
implementations -- the reference's included -- agree); the `data` field says so.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RES8 = {"pool": [4, 3], "n_feature_maps": 45, "n_layers": 6, "use_dilation": False, "n_labels": 12}
F_ALG_MODEL = 74.35e6          # FLOP / clip, conv + linear, 2 x MAC
F_ALG_FRONTEND = 1.22e6        # FLOP / clip, FFT-based count
B_ALG = 64048                  # HBM bytes / clip end to end: 16 000 fp32 samples in + 12 fp32 logits out
B_BUILT = 96368                # what kws_forward_wav's two kernels move: + the (101, 40) fp32 feature map written and re-read
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32-input matrix (= fp32 vector) peak
PEAK_BF16_MFMA_TFLOPS = 2516.0 # MI355X_MICROARCH.md: dense bf16 matrix peak (spec); 2.0 PF sustained on this box (tools/coexec_probe_bf16)
PEAK_HBM_GBS = 8000.0
EST_MS_PER_CLIP = 2.1e-4     # wav -> logits, for sizing the pre-warm only
KERNEL_SOURCE = "honk2_amd/csrc/res8_f16x3.hip"


def source_digest(rel):
    """sha256 of a source file of this tree (tools/summarize_prof.py stores the same digest in the counter summaries)."""
    import hashlib
    try:
        with open(os.path.join(ROOT, rel), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


PMC_SUMMARIES = ("profiles/r05/final_summary.json", "profiles/r04/final_summary.json", "profiles/r04/v12_summary.json", "profiles/r03/final_summary.json", "profiles/r03/v11_summary.json", "profiles/r02/final_summary.json")   # newest first


SYNTH_BLOCK = 1024


def synth_wav(torch, lo, hi, seed, device):
    """Clips [lo, hi) of the global synthetic batch, generated on the device: 0.1*randn clamped to
    [-1,1]; every 12th clip exact zeros; every 12th+1 a 0.5-amplitude 1 kHz tone (carrying the clip's noise at -37 dB as
    dither).  Clip i is a function of (seed, i) alone -- blocks of 1024 clips, one generator seed each -- so every
    sharding of the batch sees the same clips."""
    n = hi - lo
    wav = torch.empty((n, 16000), dtype=torch.float32, device=device)
    for blk in range(lo // SYNTH_BLOCK, (hi + SYNTH_BLOCK - 1) // SYNTH_BLOCK):
        g = torch.Generator(device=device).manual_seed(seed + blk)
        x = (0.1 * torch.randn((SYNTH_BLOCK, 16000), generator=g, device=device)).clamp_(-1, 1)
        a, b = max(lo, blk * SYNTH_BLOCK), min(hi, (blk + 1) * SYNTH_BLOCK)
        wav[a - lo:b - lo] = x[a - blk * SYNTH_BLOCK:b - blk * SYNTH_BLOCK]
    t = torch.arange(16000, device=device, dtype=torch.float64) / 16000.0
    tone = (0.5 * torch.sin(2 * torch.pi * 1000.0 * t)).float()
    idx = torch.arange(lo, hi, device=device)
    tones = (idx % 12) == 1
    wav[tones] = tone + 0.05 * wav[tones]
    wav[(idx % 12) == 0] = 0
    return wav


def build_model(torch, device):
    """res8 with the model's own default initialisation (torch.manual_seed(0)) and BatchNorm statistics randomised with
    generator seed 1; returns the model and its state dict as numpy arrays for the CPU leg."""
    from honk2_amd.utils import find_cls
    torch.manual_seed(0)
    model = find_cls("model.ResNet")(dict(RES8))
    g = torch.Generator().manual_seed(1)
    sd = model.state_dict()
    for k, v in sd.items():
        if k.endswith("running_mean"):
            sd[k] = 0.3 + 0.2 * torch.randn(v.shape, generator=g)
        elif k.endswith("running_var"):
            sd[k] = 0.25 + 0.5 * torch.rand(v.shape, generator=g)
    model.load_state_dict(sd)
    sd_np = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    return model.to(device).eval(), sd_np


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cgroup_cpu_quota():
    """CPUs this process may use according to its cgroup (cpu.max, v2; cfs quota, v1), or None when unlimited / unreadable."""
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),):
        try:
            with open(path) as f:
                return parse(f.read())
        except Exception:
            pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
            q, per = float(f.read()), float(g.read())
            return None if q <= 0 else q / per
    except Exception:
        return None


CPU_BLOCK = 32      # clips per work item of the CPU port: a block's activations (32 x 727 KB before the pool) stay in a core's cache


def cpu_stage_times(torch, x, sd):
    """CPU microseconds per clip of every stage of the CPU port, ONE thread, blocks of CPU_BLOCK clips as the timed leg runs them:
    the front end's numpy / scipy steps and the torch-CPU model layer by layer."""
    import numpy as np
    import scipy.fft
    from oracle import frontend, models
    win = frontend.hann_periodic(frontend.N_FFT, np.float32)
    bank = frontend.mel_filterbank().T.astype(np.float32)
    rec = {}

    def lap(name, t0):
        rec[name] = rec.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    for lo in range(0, len(x), CPU_BLOCK):
        blk = x[lo:lo + CPU_BLOCK]
        t = time.perf_counter()
        frames = frontend.frame_signal(blk)
        t = lap("reflect_pad+framing", t)
        fw = (frames * win).astype(np.float32)
        t = lap("hann_window", t)
        spec = scipy.fft.rfft(fw, axis=-1)
        t = lap("rfft_480", t)
        pw = np.abs(spec).astype(np.float32) ** 2
        t = lap("power", t)
        mel = np.matmul(pw, bank)
        t = lap("mel_matmul", t)
        out = np.array(mel, copy=True)
        pos = out > 0
        out[pos] = np.log(out[pos])
        feats = (2.0 * out).astype(np.float32)
        t = lap("log_x2", t)
        models.forward_torch("ResNet", RES8, sd, feats, timer=rec)
    return {k: round(v / len(x) * 1e6, 2) for k, v in rec.items()}


def cpu_baseline(torch, wav_sample, sd, budget_s=12.0):
    """Time the CPU oracle on host cores over a bounded sample (about 10-20 s of CPU work on all granted threads, then ~4 s on
    ONE thread).  Returns the record and the oracle's logits for the clips it evaluated (the parity check compares the GPU's
    logits with them).  `wav_sample` is ordered so that every prefix of it straddles the whole batch.
    Threads: clips are independent, so the port runs BLOCKS of CPU_BLOCK clips -- front end and model -- on a pool of N worker threads
    with torch's intra-op pool at one thread (numpy, scipy and torch release the GIL in every heavy step; the reference spreads clips
    over DataLoader worker processes the same way, data_loader/audio_data_loader.py:10-21).  Round 3's form -- 1 024-clip chunks, numpy's
    single-threaded steps, torch's intra-op threads on a 745 MB activation tensor -- did not scale (16 threads 1.0 - 1.5 x one thread:
    conv_0 + ReLU + pool alone was 450 of 680 us per clip and memory-bound); same arithmetic per clip, bit-identical logits.
    `stages` breaks the per-clip CPU time down (one thread)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import frontend, models    # bench-only use of oracle/: the checker and the timed CPU leg, never the product
    # the one-GPU box exposes every host core but grants a 16-worker share; stay inside the affinity mask, that share and the cgroup's quota
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cgroup_cpu_quota()
    threads = max(1, min(16, affinity, int(quota) if quota and quota >= 1 else 16))
    torch.set_num_threads(1)
    x = wav_sample.cpu().numpy()

    def block(blk):
        return models.forward_torch("ResNet", RES8, sd, frontend.compute_mfccs_batch(blk, "f32"))

    def run(chunk, pool):
        blocks = [chunk[i:i + CPU_BLOCK] for i in range(0, len(chunk), CPU_BLOCK)]
        outs = list(pool.map(block, blocks)) if pool is not None else [block(b) for b in blocks]
        return np.concatenate([np.asarray(o) for o in outs], 0)

    with ThreadPoolExecutor(threads) as pool:
        run(x[:CPU_BLOCK * threads], pool)          # warm-up (thread pools, allocator)
        t0 = time.perf_counter()
        run(x[:CPU_BLOCK * threads * 2], pool)
        per_clip = (time.perf_counter() - t0) / (CPU_BLOCK * threads * 2)
        n = int(min(len(x), max(256, budget_s / max(per_clip, 1e-9))))
        n = max(256, n // 256 * 256)
        t0 = time.perf_counter()
        want = run(x[:n], pool)
        dt = time.perf_counter() - t0
    # the same code on ONE thread (BASELINE.md section 4 asks for both figures): a bounded ~4 s sample
    run(x[:CPU_BLOCK], None)
    t1 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t1 < 4.0 and n1 + CPU_BLOCK <= len(x):
        run(x[n1:n1 + CPU_BLOCK], None)
        n1 += CPU_BLOCK
    dt1 = time.perf_counter() - t1
    stages = {"unit": f"CPU microseconds per clip, one thread, blocks of {CPU_BLOCK} clips", "threads_1": cpu_stage_times(torch, x[:256], sd)}
    torch.set_num_threads(threads)
    rec = {"value": n / dt, "unit": "clips/s", "cores": threads, "kind": "port",
           "sample": f"{n} of the benchmark's clips (spread over the whole batch), blocks of {CPU_BLOCK} clips on {threads} worker threads: numpy/scipy rFFT front end (fp32) + torch-CPU fp32 res8 (one intra-op thread per worker), {dt:.1f} s",
           "one_thread": {"value": n1 / dt1, "unit": "clips/s", "cores": 1, "sample": f"{n1} clips in blocks of {CPU_BLOCK}, {dt1:.1f} s"},
           "scaling_1_to_n_threads": (n / dt) / max(n1 / dt1, 1e-9),
           "stages": stages,
           "cpu_model": cpu_model_string(), "host_cores_visible": os.cpu_count(), "affinity_cores": affinity, "cgroup_cpu_quota": quota,
           "loadavg": list(os.getloadavg()) if hasattr(os, "getloadavg") else None,
           "torch_parallel_info": torch.__config__.parallel_info().strip().split("\n")[0:4]}
    return rec, want


def spread_indices(total, count):
    """`count` clip indices spread evenly over [0, total), ordered so that every prefix is itself spread over the whole range
    (the CPU leg evaluates a time-bounded prefix): the even grid is visited in bit-reversed order (van der Corput), so a prefix
    of 2^k entries is an even grid of its own and any other prefix lies between two such grids."""
    import numpy as np
    grid = (np.arange(count, dtype=np.int64) * total) // count
    bits = max(1, int(count - 1).bit_length())
    k = np.arange(1 << bits, dtype=np.int64)
    rev = np.zeros_like(k)
    for b in range(bits):
        rev |= ((k >> b) & 1) << (bits - 1 - b)
    order = rev[rev < count]
    return grid[order]


def secondary_configs(torch, device):
    """BASELINE configs[2] (res15, `bf16`, B = 4096) and configs[4] (cnn-trad-pool2, `fp16`, B = 8192) in their own dtype, with the
    models' own initialisation on the benchmark's synthetic clips: wav -> logits (front end + model, the headline's path) and
    features -> logits (the model alone), reported next to the headline and never mixed into it.  TFLOP/s are algorithmic
    (2 x MAC of the conv / linear stack), the roof is the dense 16-bit MFMA peak (single-term products); for the layer-by-layer
    res15 plan SURVEY.md 8(d)'s HBM-side roof (12.0 MB of bf16 activation traffic per clip) is quoted beside it.  `parity`
    compares the first 256 clips with the fp32 CPU oracle at the dtype's tolerance, argmax on clips
    whose oracle margin exceeds twice that tolerance."""
    import numpy as np
    from honk2_amd.utils import find_cls
    from oracle import frontend, models    # bench-only use of oracle/: the checker
    res15 = {"n_feature_maps": 45, "n_layers": 13, "use_dilation": True, "n_labels": 12}
    trad = {"time": 101, "frequency": 40, "dropout_prob": 0.5, "n_labels": 12,
            "conv_0": {"out_channels": 64, "kernel_size": [20, 8], "stride": [1, 1]}, "pool_0": {"kernel_size": [2, 2]},
            "conv_1": {"out_channels": 64, "kernel_size": [10, 4], "stride": [1, 1]}, "pool_1": {"kernel_size": [1, 1]}}
    out = []
    for tag, name, cfg, dtype, batch, mflop, tol in (("configs[2] res15 bf16", "ResNet", res15, "bf16", 4096, 1917.63, 2e-2),
                                                     ("configs[4] cnn-trad-pool2 fp16", "CNN", trad, "fp16", 8192, 192.37, 5e-3)):
        torch.manual_seed(7)
        model = find_cls(f"model.{name}")(dict(cfg, dtype=dtype))
        g = torch.Generator().manual_seed(1)
        sd = model.state_dict()
        for k, v in sd.items():
            if k.endswith("running_mean"):
                sd[k] = 0.3 + 0.2 * torch.randn(v.shape, generator=g)
            elif k.endswith("running_var"):
                sd[k] = 0.25 + 0.5 * torch.rand(v.shape, generator=g)
        model.load_state_dict(sd)
        sd_np = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
        model = model.to(device).eval()
        wav = synth_wav(torch, 0, batch, 1234, device)
        feats = model.engine().mfcc(wav)

        def timed(fn):
            # ~300 ms of untimed load first (the CPU baseline before this leaves the GPU idle; the clock governor settles over ~100 ms),
            # then ~200 ms timed
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            one = max(time.perf_counter() - t0, 1e-4)
            for _ in range(min(200, int(0.3 / one) + 1)):
                fn()
            reps = max(5, min(100, int(0.2 / one)))
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(reps):
                y = fn()
            ev[1].record()
            torch.cuda.synchronize()
            return ev[0].elapsed_time(ev[1]) / reps, y

        ms_wav, y = timed(lambda: model.forward_wav(wav))
        ms_feat, yf = timed(lambda: model(feats))
        nchk = 256
        ofeats = frontend.compute_mfccs_batch(wav[:nchk].cpu().numpy(), "f32")
        want = np.asarray(models.forward_torch(name, cfg, sd_np, ofeats))
        got = y[:nchk].cpu().numpy()
        err = np.abs(got - want)
        top = np.sort(want, axis=1)
        wide = (top[:, -1] - top[:, -2]) > 2.0 * tol
        par = {"clips": nchk, "max_abs_err": float(err.max()), "tol": tol, "argmax_checked": int(wide.sum()),
               "argmax_equal_on_checked": bool((got.argmax(1) == want.argmax(1))[wide].all()),
               "against": "fp32 CPU oracle (numpy front end + torch-CPU model) on the same clips"}
        par["pass"] = bool(np.isfinite(got).all() and par["max_abs_err"] <= tol and par["argmax_equal_on_checked"])
        rate = batch / ms_feat * 1e3
        rec = {"config": tag, "plan": model.plan_name(), "dtype": dtype, "batch": batch,
               "wav_to_logits": {"ms": ms_wav, "clips_per_s": batch / ms_wav * 1e3},
               "features_to_logits": {"ms": ms_feat, "clips_per_s": rate, "TFLOPs_alg": rate * mflop * 1e6 / 1e12,
                                      "frac_of_2516_TFLOPs": rate * mflop * 1e6 / 1e12 / PEAK_BF16_MFMA_TFLOPS},
               "parity": par, "finite": bool(torch.isfinite(y).all().item() and torch.isfinite(yf).all().item()),
               "what": "own initialisation, the benchmark's synthetic clips; wav -> logits = front end + model, features -> logits = the model alone"}
        if name == "ResNet":
            # a layer-by-layer res15 in bf16 moves (13 x 2 + 6 + 1) x 363 600 B = 12.0 MB per clip
            lw_bytes = (13 * 2 + 6 + 1) * 363600
            rec["roofline_layerwise"] = {"bound": "hbm", "bytes_per_clip": lw_bytes, "peak_clips_per_s": PEAK_HBM_GBS * 1e9 / lw_bytes,
                                         "achieved_clips_per_s": rate, "frac": rate * lw_bytes / (PEAK_HBM_GBS * 1e9),
                                         "note": "SURVEY.md 8(d) secondary model: bf16 activation traffic of a layer-by-layer res15 against 8 TB/s"}
        out.append(rec)
        del model, wav, feats, y, yf
        torch.cuda.empty_cache()
    return out


def shard_record(torch, model, device, full_rate, full_k_ms, full_f_ms, full_clips):
    """The per-GPU workload of BASELINE configs[3] at 8 GPUs (65 536 / 8 = 8 192 clips) measured on this one GPU: wav -> logits,
    100 steps after ~300 ms of load; efficiency = its clip rate over the full batch's (1.0 = an 8-GPU run would hold the 1-GPU per-clip cost)."""
    n = 8192
    wav = synth_wav(torch, 0, n, 1234, device)
    out = torch.empty((n, RES8["n_labels"]), dtype=torch.float32, device=device)
    engine = model.engine()
    for _ in range(int(math.ceil(300.0 / (n * EST_MS_PER_CLIP)))):     # ~300 ms of load first: see main()
        model.forward_wav(wav, out=out)
    torch.cuda.synchronize()
    engine.profile_enable(True)
    engine.profile_read()
    steps = 100
    t0 = time.perf_counter()
    for _ in range(steps):
        model.forward_wav(wav, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms, f_ms, calls = engine.profile_read()
    engine.profile_enable(False)
    k_ms, f_ms = k_ms / max(calls, 1), f_ms / max(calls, 1)
    rate = n * steps / dt
    return {"clips": n, "steps": steps, "ms_per_step": 1e3 * dt / steps, "clips_per_s": rate,
            "res8_kernel_ms": k_ms, "frontend_kernel_ms": f_ms,
            "efficiency_vs_full_batch": rate / full_rate,
            "res8_kernel_efficiency": (full_k_ms / full_clips) / (k_ms / n) if k_ms > 0 else None,
            "frontend_kernel_efficiency": (full_f_ms / full_clips) / (f_ms / n) if f_ms > 0 else None,
            "projection_not_a_measurement": {"eight_gpu_speedup_if_every_rank_held_this_shard_rate": 8.0 * rate / full_rate,
                                             "note": "arithmetic on ONE GPU's shard rate; no multi-GPU run is behind it -- the measured curve is the driver's SCALE record"},
            "what": "wav -> logits on 8 192 clips = one GPU's shard of the 8-GPU run (BASELINE configs[3]); the RCCL all-gather of 393 KB of logits is not in it"}


def config1_record(torch, model, sd, device, full_rate):
    """BASELINE configs[1]: res8 fp32, batch 1 024, one GPU -- wav -> logits per CALL (a batch this small is two clips per persistent
    workgroup: launch, ramp and tail effects are what it measures).  Median and mean of 200 calls after ~300 ms of load, each call bracketed by
    events on the launch stream; efficiency = its clip rate over the full batch's; parity of ALL 1 024 clips against the fp32 CPU oracle."""
    import numpy as np
    from oracle import frontend, models    # bench-only use of oracle/: the checker
    n = 1024
    wav = synth_wav(torch, 0, n, 1234, device)
    out = torch.empty((n, RES8["n_labels"]), dtype=torch.float32, device=device)
    for _ in range(int(math.ceil(300.0 / max(n * EST_MS_PER_CLIP, 0.05)))):
        model.forward_wav(wav, out=out)
    torch.cuda.synchronize()
    calls = 200
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(calls)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        model.forward_wav(wav, out=out)
        b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    med = ms[len(ms) // 2]
    want = np.asarray(models.forward_torch("ResNet", RES8, sd, frontend.compute_mfccs_batch(wav.cpu().numpy(), "f32")))
    par = parity_record(out.cpu().numpy(), want)
    par["against"] = "fp32 CPU oracle (numpy front end + torch-CPU model) on all 1 024 clips"
    return {"config": "configs[1] res8 fp32 B=1024", "plan": model.plan_name(), "dtype": "f32 (f16x3)", "batch": n,
            "wav_to_logits": {"ms_median": med, "ms_p10": ms[len(ms) // 10], "ms_p90": ms[9 * len(ms) // 10], "calls": calls,
                              "clips_per_s": n / med * 1e3, "ms_per_call_back_to_back_wall": 1e3 * wall / calls,
                              "clips_per_s_back_to_back": n * calls / wall},
            "efficiency_vs_full_batch": (n / med * 1e3) / full_rate,
            "efficiency_vs_full_batch_back_to_back": (n * calls / wall) / full_rate,
            "workgroups": {"clips_per_res8_workgroup": n / 512.0, "note": "512 persistent workgroups (2 per CU) take clips from a device-wide counter: at 1 024 clips a workgroup sees two"},
            "parity": par,
            "what": "wav -> logits on 1 024 clips per call (front end + fused res8), device-resident input; event-timed per call, and the wall time of the same 200 calls issued back to back"}


def h2d_record(torch, model, device, nclips, pcm16=False):
    """The PCIe-inclusive variant SURVEY.md 8(d) asks for beside the headline (never `value`): the waveforms start in PINNED
    host memory; chunks of 8 192 clips are copied on a second stream while the previous chunk computes.  `pcm16`: the clips
    cross the bus as the 16-bit PCM they are on disk (`kws_forward_pcm16` decodes x / 32768 in the
    front end's staging load), half the bytes per clip."""
    chunk = 8192
    nclips = nclips // chunk * chunk
    if nclips == 0:
        return None
    dtype = torch.int16 if pcm16 else torch.float32
    host = torch.empty((nclips, 16000), dtype=dtype).pin_memory()
    if pcm16:
        host.random_(-3277, 3277)
    else:
        host.normal_(0.0, 0.1)
    bufs = [torch.empty((chunk, 16000), dtype=dtype, device=device) for _ in range(2)]
    out = torch.empty((nclips, RES8["n_labels"]), dtype=torch.float32, device=device)
    copy_s, comp_s = torch.cuda.Stream(device), torch.cuda.current_stream(device)
    ready = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]

    def one_pass():
        for i in range(nclips // chunk):
            b = i & 1
            with torch.cuda.stream(copy_s):
                if i >= 2:
                    copy_s.wait_event(freed[b])
                bufs[b].copy_(host[i * chunk:(i + 1) * chunk], non_blocking=True)
                ready[b].record(copy_s)
            comp_s.wait_event(ready[b])
            model.forward_wav(bufs[b], out=out[i * chunk:(i + 1) * chunk])
            freed[b].record(comp_s)
        torch.cuda.synchronize()

    one_pass()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        one_pass()
    dt = (time.perf_counter() - t0) / reps
    host_bytes = host.element_size()
    del host, bufs
    return {"clips": nclips, "ms_per_pass": 1e3 * dt, "clips_per_s": nclips / dt,
            "h2d_GBps": nclips * 16000 * host_bytes / dt / 1e9, "input": "int16 PCM" if pcm16 else "float32",
            "what": "wav in pinned host memory -> logits on the device, copies of 8 192-clip chunks overlapped with compute (two streams); PCIe-bound, reported beside `value`, never as it"}


def literal_tone_record(torch, model, sd, device):
    """SURVEY.md 8(d)'s tone clip EXACTLY as specified -- 0.5 sin(2 pi 1000 t), no dither; every such clip of the batch is this one
    clip -- through four front ends: the default three-term fp16 DFT (kws_forward_wav, the benchmark's path), the fp32-input MFMA
    front end (KWS_FRONTEND_IMPL=fp32), the oracle's complex64-style variant and the oracle in float64; the GPU feature maps go
    through the GPU model, the oracle's through the oracle's model.  A bin-centred sine leaves every off-peak mel band pure rounding
    noise (the exact value is ~1e-14 of the peak), so the FEATURES of those bands differ between any two implementations by whole
    units; what matters for the north-star bar is what that does to the LOGITS, which this record states pair by pair."""
    import numpy as np
    from honk2_amd.utils import AudioProcessor
    from oracle import frontend, models    # bench-only use of oracle/: the checker
    t = np.arange(16000, dtype=np.float64) / 16000.0
    tone = (0.5 * np.sin(2.0 * np.pi * 1000.0 * t)).astype(np.float32)[None, :]
    wav = torch.from_numpy(tone).to(device)
    feats, logits = {}, {}
    logits["gpu_f16x3"] = model.forward_wav(wav).cpu().numpy().astype(np.float64)
    feats["gpu_f16x3"] = AudioProcessor().compute_mfccs_batch(wav).cpu().numpy()
    old = os.environ.get("KWS_FRONTEND_IMPL")
    os.environ["KWS_FRONTEND_IMPL"] = "fp32"           # read by kws_create: a fresh front-end handle takes the fp32-input MFMA kernel
    try:
        f32k = AudioProcessor().compute_mfccs_batch(wav)
    finally:
        if old is None:
            os.environ.pop("KWS_FRONTEND_IMPL", None)
        else:
            os.environ["KWS_FRONTEND_IMPL"] = old
    feats["gpu_fp32_mfma"] = f32k.cpu().numpy()
    logits["gpu_fp32_mfma"] = model(f32k).cpu().numpy().astype(np.float64)
    feats["oracle_f32"] = frontend.compute_mfccs_batch(tone, "f32")
    logits["oracle_f32"] = np.asarray(models.forward_torch("ResNet", RES8, sd, feats["oracle_f32"])).astype(np.float64)
    feats["oracle_f64"] = frontend.compute_mfccs_batch(tone, "f64")
    logits["oracle_f64"] = models.forward_numpy("ResNet", RES8, sd, feats["oracle_f64"], np.float64)
    mel = frontend.mel_power(tone, "f64")
    near = mel > 1e-4 * mel.max()                      # bands within 40 dB of the clip's strongest band
    names = list(logits)
    pairs = {}
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            df = np.abs(feats[a].astype(np.float64) - feats[b].astype(np.float64))
            pairs[f"{a} vs {b}"] = {"max_abs_dlogit": float(np.abs(logits[a] - logits[b]).max()),
                                    "argmax_equal": bool(logits[a].argmax(1)[0] == logits[b].argmax(1)[0]),
                                    "max_abs_dfeature_within_40dB_of_peak": float(df[near].max()),
                                    "max_abs_dfeature_all_bands": float(df.max())}
    top = np.sort(logits["oracle_f64"][0])
    worst = max(v["max_abs_dlogit"] for v in pairs.values())
    return {"clip": "0.5 * sin(2 pi 1000 t), 16 000 samples, float32 (SURVEY.md 8d, literal)", "pairs": pairs,
            "oracle_f64_margin_top1_top2": float(top[-1] - top[-2]), "worst_pair_max_abs_dlogit": worst,
            "tol": 1e-3, "pass": bool(worst <= 1e-3 and all(v["argmax_equal"] for v in pairs.values())),
            "bands_within_40dB": int(near.sum()), "bands_total": int(near.size),
            # how to read it: a bin-centred tone leaves ~90 % of the mel bands pure rounding noise, so `pass` (every pair within 1e-3) fails for ANY
            # two implementations, the oracle's own fp32-style and float64 variants included.  What can hold, and is asserted by the GPU test:
            "all_pairs_argmax_equal": bool(all(v["argmax_equal"] for v in pairs.values())),
            "all_pairs_agree_on_bands_within_40dB_of_peak": bool(all(v["max_abs_dfeature_within_40dB_of_peak"] < 1e-4 for v in pairs.values())),
            "gpu_default_within_the_oracles_own_f32_f64_spread": bool(pairs["gpu_f16x3 vs oracle_f32"]["max_abs_dlogit"]
                                                                      <= pairs["oracle_f32 vs oracle_f64"]["max_abs_dlogit"])}


def live_traffic(batch, kernel_substr="res8h_kernel", timeout_s=150):
    """HBM bytes per launch of the dominant kernel, measured NOW: two child runs of this very script (two timed steps, nothing else) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` -- separate passes, KB units, x2 on the gfx950 fetch counter, as
    MI355X_MICROARCH.md prescribes.  Children are started as ordinary child processes with python3 itself behind `--` (no exec from this process).
    Returns a dict, or None where rocprofv3 is not installed; an error never takes the headline line down."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="kws_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "-d", d, "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "2", "--warmup", "1", "--prewarm-ms", "0", "--batch", str(batch), "--no-cpu-baseline", "--no-secondary", "--no-shard",
                   "--no-h2d", "--no-live-traffic"]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "KWS_FORCE_DIST", "KWS_BENCH_DUMP"):
                env.pop(k, None)
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
            if r.returncode != 0:
                return {"error": f"rocprofv3 --pmc {ctr} exited with {r.returncode}: " + (r.stderr or r.stdout)[-300:]}
            rows = [row for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True) for row in csv.DictReader(open(path))
                    if row["Counter_Name"] == ctr]
            vals = [float(row["Counter_Value"]) for row in rows if kernel_substr in row["Kernel_Name"]]
            if not vals:
                return {"error": f"no {ctr} rows for {kernel_substr}"}
            got[ctr] = (sum(vals) / len(vals), len(vals))
            fe = [float(row["Counter_Value"]) for row in rows if "frontend_f16_kernel" in row["Kernel_Name"]]
            if fe:
                got["fe_" + ctr] = sum(fe) / len(fe)
        except Exception as exc:
            return {"error": repr(exc)}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {"bytes": (2.0 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024.0, "FETCH_SIZE_KB_raw": got["FETCH_SIZE"][0],
           "WRITE_SIZE_KB": got["WRITE_SIZE"][0], "launches_averaged": [got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]]}
    if "fe_FETCH_SIZE" in got and "fe_WRITE_SIZE" in got:     # the front-end kernel of the same runs (algorithmic: 64 000 B in + 16 160 B out per clip)
        out["frontend_bytes"] = (2.0 * got["fe_FETCH_SIZE"] + got["fe_WRITE_SIZE"]) * 1024.0
    return out


def parity_record(got, want, tol=1e-3):
    """GPU logits vs the oracle's on the same clips: the north-star bar (|diff| <= 1e-3, argmax equal).  With |diff| <= e on
    every logit the argmax can only differ where the oracle's own top-1 / top-2 margin is below 2 e, so a mismatch on a
    wider margin is a failure whatever the tolerance; mismatches on near-ties (two fp32 evaluations of the same
    clip do not order a 1e-6 margin reliably) are reported, not hidden."""
    import numpy as np
    err = np.abs(got - want)
    top = np.sort(want, axis=1)
    margin = top[:, -1] - top[:, -2]
    miss = got.argmax(1) != want.argmax(1)
    max_err = float(err.max())
    worst_margin = float(margin[miss].max()) if miss.any() else 0.0
    ok = bool(np.isfinite(got).all() and max_err <= tol and worst_margin <= 2.0 * max_err + 1e-7)
    return {"clips": int(len(want)), "max_abs_err": max_err, "tol": tol, "argmax_equal": bool(not miss.any()),
            "argmax_mismatches": int(miss.sum()), "mismatch_max_margin": worst_margin,
            "min_margin": float(margin.min()), "against": "cpu_baseline's oracle logits (fp32 port) on the same clips",
            "pass": ok}


def rccl_version(torch):
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:   # noqa: BLE001
        return None


def load_launcher():
    """honk2_amd/launch.py loaded BY PATH: importing the package would import torch, and the launching process stays torch-free."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_kws_launch", os.path.join(ROOT, "honk2_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prewarm-ms", type=float, default=300.0, help="untimed load before the W warm-up steps (clock governor); 0 = none")
    ap.add_argument("--batch", type=int, default=65536, help="GLOBAL batch (clips per step over all GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[2] / configs[4] records (N = 1 only)")
    ap.add_argument("--no-shard", action="store_true", help="skip the 8 192-clip shard record (N = 1 only)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the pinned-host (PCIe-inclusive) record (N = 1 only)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not re-run two short rocprofv3 --pmc passes of this command for roofline.traffic (N = 1 only)")
    ap.add_argument("--allow-fallback", action="store_true", help="N > 1: if the side-stream all-gather is refused, gather on the compute stream instead of failing the run")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit(f"--gpus {args.gpus}: need at least one GPU")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: this process becomes the launcher.  It has not imported torch, let alone touched the GPU; the N
        # ranks are FRESH children (never an exec from a process that has initialised HIP) and rank 0's JSON line reaches stdout through
        # the inherited descriptor.  The reference picks its GPU count from one integer inside one command the same way
        # (run/test.py:69-70, utils/torch_utils.py:9-22).
        raise SystemExit(load_launcher().launch_ranks(args.gpus, script=os.path.abspath(__file__), argv=sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from honk2_amd import dist_utils

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and (world > 1 or args.gpus > 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # KWS_BENCH_BACKEND / KWS_BENCH_ONE_DEVICE are rehearsal knobs only (several ranks on one GPU over gloo, to exercise
    # the N > 1 code path on a single-GPU box); the real multi-GPU run uses RCCL with one GPU per rank.
    # KWS_FORCE_DIST=1 (under torchrun, one rank): the N > 1 code path -- process group, side-stream all-gather, barrier, all-reduce --
    # on a one-rank RCCL communicator, so that a one-GPU box executes every RCCL call the multi-GPU run makes.
    backend = os.environ.get("KWS_BENCH_BACKEND", "nccl")
    rank, world = dist_utils.init_from_env(backend)
    dist_on = dist_utils.active()
    local = dist_utils.local_device_index()
    if local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local} but this node shows {torch.cuda.device_count()} GPU(s) -- --gpus {args.gpus} needs one GPU per rank")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    lo, hi = dist_utils.shard_bounds(args.batch, rank, world)
    nloc = hi - lo
    model, sd = build_model(torch, device)
    wav = synth_wav(torch, lo, hi, 1234, device)
    logits = torch.empty((nloc, RES8["n_labels"]), dtype=torch.float32, device=device)
    counts = [b - a for a, b in (dist_utils.shard_bounds(args.batch, r, world) for r in range(world))]
    gathered = torch.empty((args.batch, RES8["n_labels"]), dtype=torch.float32, device=device) if dist_on else None

    # N > 1: the all-gather of step i runs on a side stream while step i + 1 computes (two logits / gather buffers in
    # rotation; the compute stream waits for the collective that last read a buffer before overwriting it).  Everything is
    # drained inside the timed region (device-wide synchronize + barrier), so the collectives are fully paid for.
    # (KWS_BENCH_SELF_GATHER=1, one rank: the same stream / event / buffer rotation with a device copy standing in for the
    # collective -- lets a one-GPU test exercise the ordering logic that the multi-GPU run relies on)
    self_gather = world == 1 and bool(os.environ.get("KWS_BENCH_SELF_GATHER"))
    overlap = (dist_on and backend == "nccl" and len(set(counts)) == 1) or self_gather
    overlap_fallback = None
    if self_gather:
        gathered = torch.zeros_like(logits)
    lbufs = [logits, torch.empty_like(logits)] if overlap else [logits]
    gbufs = [gathered, torch.empty_like(gathered)] if overlap else [gathered]
    coll_stream = torch.cuda.Stream(device) if overlap else None
    if overlap and not self_gather:
        # One collective on the side stream before anything is timed: if this build of torch / RCCL refuses the pattern (an
        # exception every rank sees alike), fall back to the gather on the compute stream instead of losing the run.
        try:
            with torch.cuda.stream(coll_stream):
                dist.all_gather_into_tensor(gbufs[1], lbufs[1])
            torch.cuda.synchronize()
        except Exception as exc:   # noqa: BLE001
            if not args.allow_fallback:       # a run that silently measured a different collective pattern is not the run that was asked for
                raise SystemExit(f"bench: rank {rank}: the side-stream all-gather failed ({exc!r}); pass --allow-fallback to gather on the compute stream instead")
            if rank == 0:
                print(f"bench: overlapped all-gather unavailable ({exc!r}); gathering on the compute stream (--allow-fallback)", file=sys.stderr)
            overlap, overlap_fallback = False, repr(exc)
            lbufs, gbufs, coll_stream = [logits], [gathered], None
    coll_done = [None, None]
    state = {"i": 0}

    def step():
        i = state["i"]
        state["i"] = i + 1
        b = (i & 1) if overlap else 0
        if overlap and coll_done[b] is not None:
            torch.cuda.current_stream(device).wait_event(coll_done[b])
        model.forward_wav(wav, out=lbufs[b])
        if overlap:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(device))
            coll_stream.wait_event(ready)
            with torch.cuda.stream(coll_stream):
                if self_gather:
                    gbufs[b].copy_(lbufs[b])
                else:
                    dist.all_gather_into_tensor(gbufs[b], lbufs[b])   # RCCL over xGMI: the path's only collective
                coll_done[b] = torch.cuda.Event()
                coll_done[b].record(coll_stream)
        elif dist_on:
            if backend != "nccl":                                  # rehearsal over gloo: collectives on host copies
                gbufs[0].copy_(dist_utils.all_gather_rows(lbufs[0].cpu(), counts))
            else:
                gbufs[0].copy_(dist_utils.all_gather_rows(lbufs[0], counts))

    # The clock governor needs ~100 ms of load to settle: the same 8 192-clip launches average 0.44 ms each over 10 repetitions from idle
    # and 0.35 ms over 1 000 (tools/ramp_probe.sh).  A step of the N = 8 run is 1.8 ms, so W = 5 warm-up steps would leave the K timed ones on
    # the ramp; an untimed pre-warm of ~args.prewarm_ms (a step count fixed by the shard size, the same on every rank) comes first.
    prewarm_steps = min(400, int(math.ceil(args.prewarm_ms / max(nloc * EST_MS_PER_CLIP, 1e-3)))) if args.prewarm_ms > 0 else 0
    for _ in range(prewarm_steps):
        step()
    for _ in range(args.warmup):
        step()
    engine = model.engine()
    torch.cuda.synchronize()
    engine.profile_enable(True)
    engine.profile_read()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    model_ms, front_ms, calls = engine.profile_read()
    engine.profile_enable(False)
    last = ((state["i"] - 1) & 1) if overlap else 0
    logits, gathered = lbufs[last], gbufs[last]

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    if not bool(torch.isfinite(logits).all()):
        raise SystemExit("non-finite logits")
    rank_devices, gather_ok = None, None
    if dist_on:
        # which device each rank computed on (did RCCL see N ranks on N GPUs: answerable from the line), and rank r's slice of the gathered
        # logits against what rank r computed itself in the last step
        mine = {"rank": rank, "local_device": local, "name": torch.cuda.get_device_name(local), "pci_bus_id": getattr(torch.cuda.get_device_properties(local), "pci_bus_id", None)}
        rank_devices = [None] * dist.get_world_size()
        dist.all_gather_object(rank_devices, mine)
        gather_ok = bool(torch.equal(gathered[lo:hi], logits))
        flag = torch.tensor([0 if gather_ok else 1], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            raise SystemExit(f"rank {rank}: the gathered logits differ from a rank's own shard")

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
        frontend_traffic_live = None
        roofline = {"bound": "mfma", "kernel": kern, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                    "frac": achieved / peak, "traffic": None, "kernel_ms": k_ms, "launches": calls,
                    "flop_per_launch": F_ALG_MODEL * nloc, "note": note}
        # HBM bytes per launch cannot be measured from inside the process; when the committed counter summary of this very
        # workload (same kernel, same clips per launch) is present, quote it: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate passes, KB units, x2 correction on the gfx950 fetch counter (MI355X_MICROARCH.md).
        # (a STATIC figure: it describes the committed profile, not this very run)
        if plan == "res8_fused" and nloc == 65536:
            for rel in PMC_SUMMARIES:
                pmc = os.path.join(ROOT, rel)
                try:
                    with open(pmc) as f:
                        summ = json.load(f)
                    ent = next(v for k, v in summ.items() if "res8h_kernel" in k)
                    roofline["traffic"] = (2.0 * ent["FETCH_SIZE"] + ent["WRITE_SIZE"]) * 1024.0
                    roofline["traffic_source"] = f"static: {rel} (rocprofv3 --pmc of this command, committed; not measured in this run)"
                    # the summary names the kernel source it profiled: a kernel edited since then is flagged, not quoted silently
                    profiled = summ.get("_sources", {}).get(KERNEL_SOURCE)
                    roofline["traffic_profile_matches_kernel_source"] = (profiled == source_digest(KERNEL_SOURCE)) if profiled else None
                    break
                except Exception:
                    continue
            # ... and measured in THIS run when rocprofv3 is here: two short child runs of this command under --pmc (after the timed region: they
            # cannot disturb it).  The committed figure stays beside it, so a traffic regression shows as a disagreement inside one line.
            if world == 1 and not args.no_live_traffic:
                lt = live_traffic(args.batch)
                if lt and "bytes" in lt:
                    roofline["traffic_static_committed"] = roofline.get("traffic")
                    roofline["traffic"] = lt["bytes"]
                    roofline["traffic_live"] = lt
                    if "frontend_bytes" in lt:
                        frontend_traffic_live = lt["frontend_bytes"]
                    roofline["traffic_source"] = ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of this command, run as child processes "
                                                  "after the timed region; committed summary beside it in traffic_static_committed: " + str(roofline.get("traffic_source")))
                elif lt:
                    roofline["traffic_live"] = lt
        out = {
            "metric": "1s-clips/sec end-to-end (wav->logits), res8 GSCv2", "value": clips_per_s, "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm": {"steps": prewarm_steps, "target_ms": args.prewarm_ms, "what": "untimed steps before the W warm-up steps: the clock governor settles over ~100 ms of load"},
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": ("f32 (f16x3: fp32-accurate 3-term fp16 products, fp32 accumulate, in the conv stack and in the front end's DFT)" if plan == "res8_fused" else "f32 (bf16x6: fp32-accurate 6-term bf16 products, fp32 accumulate; fp32 front end)" if plan == "res8_fused_bf16x6" else "f32"), "data": "synthetic (SURVEY.md 8d inputs; deviation: the 1 kHz tone clips carry their clip's noise at -37 dB as dither)",
            "config": {"workload": f"res8 fp32 wav->logits, global batch {args.batch} one-second 16 kHz clips "
                                   f"(BASELINE configs[3]), {nloc} clips/GPU, random-init weights",
                       "global_batch": args.batch, "clips_per_gpu": nloc, "n_samples": 16000,
                       "plan": model.plan_name(), "parallelism": f"dp{world} (clip sharding, logits all-gather" + (" on a side stream, overlapped with the next step" if overlap else "") + ")"},
            "collective": ({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "forced_one_rank_group": dist_utils.forced() and world == 1,
                            "rccl_version": rccl_version(torch) if backend == "nccl" else None,
                            "clips_per_gpu": counts, "devices": rank_devices,
                            "op": "all_gather_into_tensor of the (B/N, 12) fp32 logits per step", "overlapped_on_side_stream": bool(overlap),
                            "overlap_fallback": overlap_fallback,
                            "gathered_equals_local_shard_on_rank_0": gather_ok} if dist_on else None),
            "roofline": roofline,
            "frontend": {"kernel": "frontend_f16_kernel (reflect pad + Hann + 480-point DFT as three-term fp16 MFMA products + mel + log)",
                         "kernel_ms": f_ms, "bound": "hbm", "bytes_per_clip": 80160, "traffic": frontend_traffic_live,
                         "hbm_GBps_algorithmic": (80160 * nloc / (f_ms * 1e-3) / 1e9) if f_ms > 0 else 0.0,
                         "frac_of_8TBps": (80160 * nloc / (f_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if f_ms > 0 else 0.0},
            "b_alg": {"end_to_end": B_ALG, "as_built": B_BUILT,
                      "note": "kws_forward_wav runs two kernels: the (101,40) fp32 feature map is written to the workspace and re-read (+2 x 16 160 B/clip)"},
            "end_to_end": {"hbm_frac_of_8TBps": clips_per_s / world * B_ALG / (PEAK_HBM_GBS * 1e9),
                           "hbm_frac_of_8TBps_as_built": clips_per_s / world * B_BUILT / (PEAK_HBM_GBS * 1e9),
                           "flop_frac_of_fp32_peak": clips_per_s / world * (F_ALG_MODEL + F_ALG_FRONTEND) / (PEAK_F32_MFMA_TFLOPS * 1e12)},
        }
        failed = False
        if world == 1 and not args.no_cpu_baseline:
            # the CPU leg evaluates a time-bounded prefix of a sample that is spread over the WHOLE batch (every 4th clip, in an
            # order whose prefixes are spread too), and the parity check covers exactly those clips
            idx = torch.from_numpy(spread_indices(nloc, min(16384, nloc))).to(device)
            out["cpu_baseline"], want = cpu_baseline(torch, wav[idx], sd)
            out["parity"] = parity_record(logits[idx[:len(want)]].cpu().numpy(), want)
            out["parity"]["clip_index_range"] = [int(idx[:len(want)].min()), int(idx[:len(want)].max())]
            failed = not out["parity"]["pass"]
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["parity_literal_tone"] = literal_tone_record(torch, model, sd, device)
            except Exception as exc:                   # never let an extra record take the headline line down
                out["parity_literal_tone"] = {"error": repr(exc)}
        if world == 1 and not args.no_shard and args.batch >= 16384:
            out["shard"] = shard_record(torch, model, device, clips_per_s, k_ms, f_ms, nloc)
        if world == 1 and not args.no_h2d:
            try:
                out["h2d_inclusive"] = h2d_record(torch, model, device, min(args.batch, 32768))
                out["h2d_inclusive_pcm16"] = h2d_record(torch, model, device, min(args.batch, 32768), pcm16=True)
            except Exception as exc:
                out["h2d_inclusive"] = {"error": repr(exc)}
        if world == 1 and not args.no_secondary:
            try:
                out["secondary"] = secondary_configs(torch, device)
                out["secondary"].append(config1_record(torch, model, sd, device, clips_per_s))
                failed = failed or not all(r["parity"]["pass"] for r in out["secondary"])
            except Exception as exc:                   # never let an extra record take the headline line down
                out["secondary"] = {"error": repr(exc)}
        dump = os.environ.get("KWS_BENCH_DUMP")        # tests: the (gathered) logits of the last step, for comparison across N
        if dump:
            import numpy as np
            np.save(dump, (gathered if (dist_on or self_gather) else logits).cpu().numpy())
        print(json.dumps(out))
        if failed:
            raise SystemExit("parity check failed: " + json.dumps({"headline": out.get("parity"), "secondary": [r.get("parity") for r in out.get("secondary", []) if isinstance(r, dict)]}))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
