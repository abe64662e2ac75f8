#!/usr/bin/env python3
"""Phase timeline of conv3x3_triple_kernel's workgroups (needs a -DT3_TIMING build, KWS_T3_TIMING=<file> and KWS_T3_TIMING_LAYER=<first layer of the run>
while tools/t3_run.py runs res15 in a 16-bit dtype)."""
import sys, numpy as np
z = np.fromfile(sys.argv[1], dtype=np.uint64).astype(np.int64).reshape(-1, 4, 12)
z = z[z[:, 0, 0] > 0]
t = z[:, :, :11] * 0.01                      # us
names = ['decode+stage issue', 'barrier', 'k-loop 1', 'epilogue 1', 'barrier', 'k-loop 2', 'epilogue 2', 'barrier', 'k-loop 3', 'epilogue 3']
d = np.diff(t, axis=2)
for wv in range(4):
    print(f'wave {wv}: ' + ' '.join(f'{n}={d[:, wv, i].mean():.2f}' for i, n in enumerate(names)), ' total %.2f us' % (t[:, wv, 10] - t[:, wv, 0]).mean())
print('workgroups', len(z), ' launch span %.1f us' % (t[:, :, 10].max() - t[:, :, 0].min()))
