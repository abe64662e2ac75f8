#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
V=$PWD/honk2_amd/variants
KWS_LIB=$V/lib_timing.so timeout -k 10 180 python tools/r8_phases.py > gpurun_out/r2_r8_phases3.log 2>&1 || { tail -5 gpurun_out/r2_r8_phases3.log; exit 1; }
grep -v amdgpu gpurun_out/r2_r8_phases3.log | tail -16
KWS_LIB=$V/lib_timing.so R8_TAG=timing timeout -k 10 120 python tools/r8_time.py 2>/dev/null
