#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
V=$PWD/honk2_amd/variants
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "model_logits or reduced or bf16 or fp16 or tiled or chunk or range_guard or alternative or larger_batch or entry_point" > gpurun_out/r2_tests_18.log 2>&1
rc=$?; tail -3 gpurun_out/r2_tests_18.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; grep -n "Error\|assert" gpurun_out/r2_tests_18.log | head; exit $rc; fi
timeout -k 10 600 python tools/bench_models.py resnet__res15 resnet__res26 resnet__res15_narrow resnet__res26_narrow resnet__res8_narrow 2>/dev/null | cut -c1-200
KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c1-200
KWS_LIB=$V/lib_t3timing.so KWS_T3_TIMING=$PWD/gpurun_out/t3_ts.bin KWS_BENCH_BATCH=1024 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>&1 | tail -1 | cut -c1-120
python3 tools/t3_phases.py gpurun_out/t3_ts.bin | head -3
