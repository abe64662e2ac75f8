#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
V=$PWD/honk2_amd/variants
KWS_LIB=$V/lib_timing.so timeout -k 10 180 python tools/r8_phases.py > gpurun_out/r2_r8_phases6.log 2>&1 || { tail -5 gpurun_out/r2_r8_phases6.log; exit 1; }
grep -v amdgpu gpurun_out/r2_r8_phases6.log | tail -4
