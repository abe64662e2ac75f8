cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python tools/stream_debug.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "streams_are_bit_identical" > gpurun_out/stream_tests.log 2>&1
rc=$?; tail -5 gpurun_out/stream_tests.log
[ $rc -ne 0 ] && exit $rc
KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c1-200
KWS_T3_STREAM=0 KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c1-200
