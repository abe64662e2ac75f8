#!/bin/bash
# round 4: clips per launch of the cnn-* plans (KWS_CNN_CHUNK) + conv_band staging (bandbase = before)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "cnn or golden or reference or sweep or neighbours" > gpurun_out/r4/cnn_tests.txt 2>&1
rc=$?; tail -5 gpurun_out/r4/cnn_tests.txt; [ $rc -eq 0 ] || exit $rc
{
for c in 1024 768 1536 512 768; do
  echo "{\"KWS_CNN_CHUNK\": $c}"
  KWS_CNN_CHUNK=$c KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 2>/dev/null | cut -c1-150
  KWS_CNN_CHUNK=$c KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 cnn__cnn-tstride4 cnn__cnn-one-fpool3 cnn__cnn-tpool2 2>/dev/null | cut -c1-150
done
echo '{"lib": "bandbase (chunk 1024, staging four loads per pass)"}'
KWS_LIB=$PWD/honk2_amd/variants/lib_bandbase.so KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 2>/dev/null | cut -c1-150
KWS_LIB=$PWD/honk2_amd/variants/lib_bandbase.so KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 cnn__cnn-tstride4 cnn__cnn-one-fpool3 cnn__cnn-tpool2 2>/dev/null | cut -c1-150
} | tee gpurun_out/r4/cnn_chunk.txt
