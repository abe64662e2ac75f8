#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
V=$PWD/honk2_amd/variants
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "res8 or wav_to_logits or range_guard or full_batch" > gpurun_out/r2_tests_4.log 2>&1
rc=$?; tail -3 gpurun_out/r2_tests_4.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
for round in 1 2 3; do
  for lib in old default prio1; do
    if [ $lib = default ]; then unset KWS_LIB; else export KWS_LIB=$V/lib_$lib.so; fi
    R8_TAG=$lib timeout -k 10 120 python tools/r8_time.py 2>/dev/null >> gpurun_out/r2_r8_ab4.log || exit 1
  done
done
unset KWS_LIB
cat gpurun_out/r2_r8_ab4.log
KWS_LIB=$V/lib_timing.so timeout -k 10 180 python tools/r8_phases.py > gpurun_out/r2_r8_phases4.log 2>&1 || { tail -5 gpurun_out/r2_r8_phases4.log; exit 1; }
grep -v amdgpu gpurun_out/r2_r8_phases4.log | tail -4
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_4.log 2>&1 || { echo bench failed; tail -5 gpurun_out/r2_bench_4.log; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r2_bench_4.log') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['frontend']['kernel_ms'])
PY
