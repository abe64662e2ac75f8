#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
V=$PWD/honk2_amd/variants
for dbg in 0 4 8 12 0; do
  KWS_R8_DEBUG=$dbg KWS_LIB=$V/lib_ablate.so R8_TAG=ablate timeout -k 10 120 python tools/r8_time.py 2>/dev/null || exit 1
  KWS_R8_DEBUG=$dbg KWS_LIB=$V/lib_ablate_t.so timeout -k 10 180 python tools/r8_phases.py > gpurun_out/r2_r8_phases9.log 2>&1 || { tail -5 gpurun_out/r2_r8_phases9.log; exit 1; }
  python3 - <<'PY'
import numpy as np
z=np.load('gpurun_out/r8_clip_times.npz'); rt=z['rt']; ts=z['ts']
tot=ts[:,0,7]-ts[:,0,0]; us=(rt[:,1]-rt[:,0])*0.01
print('   timing build: mean ticks %.0f  mean us %.1f  clock %.3f GHz' % (tot.mean(), us.mean(), (tot/us/1e3).mean()))
PY
done
