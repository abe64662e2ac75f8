#!/bin/bash
# round 4: cnn-* iteration -- parity tests of the cnn plans, then new / variants alternating on cnn-trad-pool2 fp16 + f32 (CNN_VARIANTS = names under honk2_amd/variants)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "cnn or golden or reference or sweep or neighbours" > gpurun_out/r4/cnn_tests.txt 2>&1
rc=$?; tail -5 gpurun_out/r4/cnn_tests.txt; [ $rc -eq 0 ] || exit $rc
{
for rep in 1 2; do
  for v in new $CNN_VARIANTS; do
    lib=$PWD/honk2_amd/variants/lib_$v.so; [ $v = new ] && lib=$PWD/honk2_amd/libkws_hip.so
    echo "{\"variant\": \"$v\"}"
    KWS_LIB=$lib KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 2>/dev/null | cut -c1-150
    KWS_LIB=$lib KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 cnn__cnn-tstride4 2>/dev/null | cut -c1-150
  done
done
} | tee gpurun_out/r4/cnn_iter.txt
