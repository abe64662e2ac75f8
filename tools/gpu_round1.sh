#!/bin/bash
# first GPU call of round 2: tests, bench, res8 ablations + phase timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r2_tests_1.log 2>&1
rc=$?
tail -5 gpurun_out/r2_tests_1.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r2_bench_1.log 2>&1 || { echo bench failed; tail -5 gpurun_out/r2_bench_1.log; exit 1; }
tail -c 1500 gpurun_out/r2_bench_1.log
for dbg in 0 1 2 3; do
  KWS_R8_DEBUG=$dbg timeout -k 10 120 python tools/r8_time.py >> gpurun_out/r2_r8_ablate.log 2>&1 || exit 1
done
KWS_R8_WGS_PER_CU=1 timeout -k 10 120 python tools/r8_time.py >> gpurun_out/r2_r8_ablate.log 2>&1 || exit 1
KWS_LIB=$PWD/honk2_amd/variants/lib_timing.so timeout -k 10 120 python tools/r8_phases.py > gpurun_out/r2_r8_phases.log 2>&1 || exit 1
cat gpurun_out/r2_r8_ablate.log
