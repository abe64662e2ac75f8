#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "model_logits or reduced or bf16 or fp16 or tiled or layerwise or chunk or range_guard or alternative or larger_batch or entry_point" > gpurun_out/r2_tests_12.log 2>&1
rc=$?; tail -4 gpurun_out/r2_tests_12.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; grep -n "Error\|assert" gpurun_out/r2_tests_12.log | head; exit $rc; fi
timeout -k 10 600 python tools/bench_models.py resnet__res15 resnet__res26 resnet__res15_narrow resnet__res26_narrow resnet__res8_narrow cnn__cnn-trad-pool2 cnn__cnn-tpool2 cnn__cnn-one-fstride4 2>/dev/null | cut -c1-210
KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c1-210
KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 2>/dev/null | cut -c1-210
