#!/usr/bin/env python3
"""Phase timeline of conv_cols_kernel's workgroups (needs a -DCOLS_TIMING build and KWS_BAND_TIMING=<file> while cnn-trad-pool2 `fp16` runs):
per unit k-loop / wait at the barrier behind it / epilogue (+ next image in flight) / wait for the image; then the units of the two workgroups of one CU side by side."""
import sys
import numpy as np
z = np.fromfile(sys.argv[1], dtype=np.uint64).astype(np.int64)[:512 * 4 * 8 * 8].reshape(512, 4, 8, 8)
ok = z[:, 0, 0, 0] > 0
z = z[ok]
t = z[..., :5] * 0.01                      # us
d = np.diff(t, axis=3)
names = ['k-loop', 'barrier', 'epilogue', 'image wait']
for wv in range(4):
    print(f'wave {wv}: ' + ' '.join(f'{n}={d[:, wv, :, i].mean():.2f}' for i, n in enumerate(names)), ' unit %.2f us' % (t[:, wv, :, 4] - t[:, wv, :, 0]).mean())
print('workgroups', len(z), ' span %.1f us' % (t[..., 4].max() - t[..., 0].min()))
hw = z[:, 0, 0, 5]
key = ((hw >> 32) << 32) | (hw & 0xff00)
t0 = t[..., 0].min()
for k in np.unique(key)[:2]:
    idx = np.flatnonzero(key == k)
    print('CU', hex(int(k)), 'workgroups', len(idx))
    for i in idx[:3]:
        print('   wg', i, ' '.join('[%.1f %.1f %.1f %.1f %.1f]' % tuple(t[i, 0, u] - t0) for u in range(4)))
