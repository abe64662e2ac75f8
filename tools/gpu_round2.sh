#!/bin/bash
# GPU call 2: probe of the accumulator file, res8 A/B variants (interleaved rounds), phase timeline with the realtime clock
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 120 ./tools/mfma_acc_probe > gpurun_out/r2_acc_probe.log 2>&1 || { echo probe failed; exit 1; }
cat gpurun_out/r2_acc_probe.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "res8 or wav_to_logits or range_guard or full_batch" > gpurun_out/r2_tests_2.log 2>&1
rc=$?; tail -3 gpurun_out/r2_tests_2.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
V=$PWD/honk2_amd/variants
for round in 1 2 3; do
  for lib in old default prio1 prio3 nopf; do
    if [ $lib = default ]; then unset KWS_LIB; else export KWS_LIB=$V/lib_$lib.so; fi
    R8_TAG=$lib timeout -k 10 120 python tools/r8_time.py 2>/dev/null >> gpurun_out/r2_r8_ab2.log || exit 1
  done
done
unset KWS_LIB
cat gpurun_out/r2_r8_ab2.log
KWS_LIB=$V/lib_timing.so timeout -k 10 120 python tools/r8_phases.py > gpurun_out/r2_r8_phases2.log 2>&1 || exit 1
KWS_LIB=$V/lib_timing_prio3.so timeout -k 10 120 python tools/r8_phases.py > gpurun_out/r2_r8_phases2_prio3.log 2>&1 || exit 1
tail -8 gpurun_out/r2_r8_phases2.log
