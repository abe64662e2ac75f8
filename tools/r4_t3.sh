#!/bin/bash
# round 4: tiled / pair / triple 3x3 kernels -- bit-identity and golden tests, then res15 bf16 / res26 fp16 new against variants (T3_VARIANTS), phase stamps of the triple at layer 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "triples or layer_pairs or golden or reference or reduced or neighbours or range_guard or sweep" > gpurun_out/r4/t3_tests.txt 2>&1
rc=$?; tail -5 gpurun_out/r4/t3_tests.txt; [ $rc -eq 0 ] || exit $rc
{
for rep in 1 2; do
  for v in new $T3_VARIANTS; do
    lib=$PWD/honk2_amd/variants/lib_$v.so; [ $v = new ] && lib=$PWD/honk2_amd/libkws_hip.so
    echo "{\"variant\": \"$v\"}"
    KWS_LIB=$lib KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c1-150
    KWS_LIB=$lib KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res26 2>/dev/null | cut -c1-150
  done
done
if [ -f honk2_amd/variants/lib_t3ts.so ]; then
  DT=bf16 KWS_T3_TIMING_LAYER=4 KWS_T3_TIMING=gpurun_out/t3x_4.bin KWS_LIB=$PWD/honk2_amd/variants/lib_t3ts.so timeout -k 10 200 python tools/t3_run.py && python tools/t3x_phases.py gpurun_out/t3x_4.bin
fi
} | tee gpurun_out/r4/t3_iter.txt
