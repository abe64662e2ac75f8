#!/usr/bin/env python3
"""Phase timeline of the tiled 3x3 kernel's workgroups (needs a -DT3_TIMING build and KWS_T3_TIMING=<file> while a model runs)."""
import sys, numpy as np
z = np.fromfile(sys.argv[1], dtype=np.uint64).astype(np.int64).reshape(-1, 4, 8)
ok = z[:, 0, 0] > 0
z = z[ok]
t = z[:, :, :7] * 0.01                      # us
names = ['setup', 'stage loads+LDS write', 'barrier', 'k-loop', 'epilogue', 'store drain']
d = np.diff(t, axis=2)
print('workgroups', len(z), ' per-phase mean us (wave 0):', ' '.join(f'{n}={d[:, 0, i].mean():.2f}' for i, n in enumerate(names)), ' total %.2f' % (t[:, 0, 6] - t[:, 0, 0]).mean())
print('setup split (wave 0): start->args %.2f  args->positions %.2f  positions->residual issued %.2f' % ((z[:,1,1]-z[:,0,0]).mean()*0.01, (z[:,1,0]-z[:,1,1]).mean()*0.01, (z[:,0,1]-z[:,1,0]).mean()*0.01))
hw = z[:, 0, 7]
key = ((hw >> 32) << 32) | (hw & 0xff00)      # XCC + SE/SH/CU
order = np.argsort(t[:, 0, 0])
start, end = t[:, 0, 0], t[:, 0, 6]
print('launch span us %.1f' % (end.max() - start.min()))
# per CU: gaps between the end of a workgroup and the start of the next one in the same slot
for k in np.unique(key)[:3]:
    idx = np.flatnonzero(key == k)
    idx = idx[np.argsort(start[idx])]
    print('CU', hex(int(k)), 'workgroups', len(idx))
    for i in idx[:12]:
        print('   start %.2f  end %.2f  phases %s' % (start[i] - start.min(), end[i] - start.min(), ' '.join(f'{x:.2f}' for x in d[i, 0])))
