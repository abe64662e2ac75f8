#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
V=$PWD/honk2_amd/variants
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "frontend or wav_to_logits or pcm16 or streaming or data_loader or entry_point" > gpurun_out/r2_tests_5.log 2>&1
rc=$?; tail -3 gpurun_out/r2_tests_5.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
for round in 1 2 3; do
  for lib in feold default fenoslp; do
    if [ $lib = default ]; then unset KWS_LIB; else export KWS_LIB=$V/lib_$lib.so; fi
    FE_TAG=$lib timeout -k 10 120 python tools/fe_time.py 2>/dev/null >> gpurun_out/r2_fe_ab5.log || exit 1
  done
done
unset KWS_LIB
cat gpurun_out/r2_fe_ab5.log
KWS_LIB=$V/lib_fetiming.so timeout -k 10 180 python tools/fe_phases.py > gpurun_out/r2_fe_phases5.log 2>&1 || { tail -5 gpurun_out/r2_fe_phases5.log; exit 1; }
grep -v amdgpu gpurun_out/r2_fe_phases5.log | tail -16
