#!/usr/bin/env python3
"""Phase timeline of conv_band_kernel's workgroups (needs a -DBAND_TIMING build and KWS_BAND_TIMING=<file> while a cnn-* model runs)."""
import sys, numpy as np
z = np.fromfile(sys.argv[1], dtype=np.uint64).astype(np.int64).reshape(-1, 4, 8)
z = z[z[:, 0, 0] > 0]
t = z[:, :, :5] * 0.01                      # us
names = ['stage', 'barrier', 'k-loop', 'epilogue']
d = np.diff(t, axis=2)
for wv in range(4):
    print(f'wave {wv}: ' + ' '.join(f'{n}={d[:, wv, i].mean():.2f}' for i, n in enumerate(names)), ' total %.2f us' % (t[:, wv, 4] - t[:, wv, 0]).mean())
print('workgroups', len(z), ' launch span %.1f us' % (t[:, :, 4].max() - t[:, :, 0].min()))
hw = z[:, 0, 5]
key = ((hw >> 32) << 32) | (hw & 0xff00)
start, end = t[:, 0, 0], t[:, :, 4].max(axis=1)
for k in np.unique(key)[:2]:
    idx = np.flatnonzero(key == k)
    idx = idx[np.argsort(start[idx])]
    print('CU', hex(int(k)), 'workgroups', len(idx))
    for i in idx[:10]:
        print('   start %.2f  end %.2f  wave0 phases %s' % (start[i] - start.min(), end[i] - start.min(), ' '.join(f'{x:.2f}' for x in d[i, 0])))
