"""Phase timestamps (s_memtime) of the fused res8 kernel's workgroups.  Needs a library whose res8_f16x3.hip was built
with -DR8H_TIMING: the kernel then parks sixteen timestamps per wave in the (consumed) feature rows of each clip
(0..7: clip phases; 8..15: inside layer 2 -- start, after k-step 0, after k-step 6, k-loop end, epilogue math done, past the
first barrier, map stored, past the second barrier)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from conftest import load_golden_model
from honk2_amd.utils import find_cls
tag, name, cfg, sd, feats, z = load_golden_model("model_resnet__res8.npz")
model = find_cls(f"model.{name}")(dict(cfg))
model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
model = model.to("cuda:0").eval()
B = 65536
x = (0.65 + 2.5 * torch.randn(B, 101, 40, device='cuda')).contiguous()
model(x); torch.cuda.synchronize()
x2 = (0.65 + 2.5 * torch.randn(B, 101, 40, device='cuda')).contiguous()
model(x2); torch.cuda.synchronize()
f = x2.cpu().numpy().reshape(B, -1)
names = ['stage', 'conv0', 'guard+map', 'layer0', 'layer1', 'layers2-4', 'layer5+tail']
lnames = ['kstep0', 'ksteps1-6', 'ksteps7-13', 'epi-math', 'barrier1', 'stores', 'barrier2']
for clip in (5, 20000, 40000, 65000):
    for w in range(4):
        ts = f[clip, 32 * w: 32 * w + 32].view(np.uint64).astype(np.int64)
        d = np.diff(ts[:8])
        print(clip, w, ' '.join(f'{n}={int(v)}' for n, v in zip(names, d)), 'total', int(ts[7] - ts[0]))
        print(clip, w, '   layer 2:', ' '.join(f'{n}={int(v)}' for n, v in zip(lnames, np.diff(ts[8:16]))), 'total', int(ts[15] - ts[8]))
    rt = f[clip, 128:132].view(np.uint64).astype(np.int64)
    ts0 = f[clip, 0:16].view(np.uint64).astype(np.int64)
    print(clip, 'realtime ticks (100 MHz)', int(rt[1] - rt[0]), '-> shader clock', round((ts0[7] - ts0[0]) / max(int(rt[1] - rt[0]), 1) * 0.1, 3), 'GHz')

# ---- whole-launch statistics from the 100 MHz wall clock stamps (wave 0 of every clip); a workgroup is identified by the
#      hardware slot (XCC_ID, HW_ID) of its wave 0, which is constant over the launch
rt = f[:, 128:132].copy().view(np.uint64).astype(np.int64)          # (B, 2): start, end of each clip in 10 ns ticks
hw = f[:, 132:134].copy().view(np.uint64).astype(np.int64)[:, 0]
ts_all = f[:, 0:128].copy().view(np.uint64).astype(np.int64).reshape(B, 4, 16)
np.savez_compressed('gpurun_out/r8_clip_times.npz', rt=rt, hw=hw, ts=ts_all)
dur = (rt[:, 1] - rt[:, 0]) * 0.01                                 # us
keys, wg = np.unique(hw, return_inverse=True)
G = len(keys)
span = np.array([(rt[wg == b, 1].max() - rt[wg == b, 0].min()) * 0.01 for b in range(G)])
nclips = np.bincount(wg)
t_all = (rt[:, 1].max() - rt[:, 0].min()) * 0.01
lt = np.diff(ts_all[:, 0, 8:16], axis=1)
print('layer 2, wave 0, median over clips:', ' '.join(f'{n}={int(v)}' for n, v in zip(lnames, np.median(lt, axis=0))), 'total', int(np.median(ts_all[:, 0, 15] - ts_all[:, 0, 8])))
ph = np.diff(ts_all[:, 0, 0:8], axis=1)
print('clip phases, wave 0, median over clips:', ' '.join(f'{n}={int(v)}' for n, v in zip(names, np.median(ph, axis=0))), 'total', int(np.median(ts_all[:, 0, 7] - ts_all[:, 0, 0])))
print('workgroups seen', G, 'clips per workgroup: min %d mean %.1f max %d' % (nclips.min(), nclips.mean(), nclips.max()))
print('per-clip us: mean %.1f  p5 %.1f  p50 %.1f  p95 %.1f  max %.1f' % (dur.mean(), *np.percentile(dur, [5, 50, 95]), dur.max()))
print('per-workgroup span us: min %.0f  mean %.0f  max %.0f ; launch span %.0f us' % (span.min(), span.mean(), span.max(), t_all))
