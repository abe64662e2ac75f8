#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
V=$PWD/honk2_amd/variants
for lib in timing_old timing timing_old timing; do
  KWS_LIB=$V/lib_$lib.so timeout -k 10 180 python tools/r8_phases.py > gpurun_out/r2_r8_phases8_$lib.log 2>&1 || { tail -5 gpurun_out/r2_r8_phases8_$lib.log; exit 1; }
  cp gpurun_out/r8_clip_times.npz gpurun_out/r8_clip_times_$lib.npz
  echo $lib; grep -v amdgpu gpurun_out/r2_r8_phases8_$lib.log | tail -2
  python3 - <<'PY'
import numpy as np
z=np.load('gpurun_out/r8_clip_times.npz'); rt=z['rt']; hw=z['hw']; ts=z['ts']
tot=ts[:,0,7]-ts[:,0,0]; us=(rt[:,1]-rt[:,0])*0.01
print('   mean ticks %.0f  mean us %.1f  clock %.3f GHz' % (tot.mean(), us.mean(), (tot/us/1e3).mean()))
PY
done
