#!/bin/bash
# conv3x3_stream.hip on the GPU box, one driver (run from the repo root through gpurun):  tools/stream.sh <what> [...]
#   debug    stream vs tile kernels on small models (which run shape differs, where)          -> stdout
#   test     the stream bit-identity tests                                                      -> gpurun_out/stream_tests.log
#   bench    res15 bf16 at B = 4 096: default plan, tile kernels only, every run as a stream    -> stdout
#   prof     rocprofv3 --kernel-trace --stats of the default plan                              -> gpurun_out/sprof/
#   pmc      wave-time breakdown counters (SQ_WAIT_* / SQ_ACTIVE_*; KWS_T3_STREAM=1)            -> stdout
#   timing   per-wave work / barrier cycles (needs honk2_amd/variants/lib_stream_t.so: tools/variant.sh stream_t conv3x3_stream.hip -DSTREAM_TIMING)
#   ablate   variants lib_stream_a{1,2,16,32}.so (tools/variant.sh stream_aN conv3x3_stream.hip -DSTREAM_ABLATE=N), every run as a stream
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bm() { KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=${B:-4096} timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c1-190; }
for what in "$@"; do
  case $what in
    debug) timeout -k 10 300 python tools/stream_debug.py 2>&1 | grep -v amdgpu.ids ;;
    test) timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "streams_are_bit_identical" > gpurun_out/stream_tests.log 2>&1; rc=$?; tail -4 gpurun_out/stream_tests.log; [ $rc -ne 0 ] && exit $rc ;;
    bench) echo "default (every run a stream):"; bm; echo "tile kernels only:"; KWS_T3_STREAM=0 bm; echo "odd-first runs only as streams:"; KWS_T3_STREAM=3 bm ;;
    prof) rm -rf gpurun_out/sprof
          KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/sprof --output-format csv -- python3 tools/bench_models.py resnet__res15 > gpurun_out/sprof.log 2>&1
          f=$(find gpurun_out/sprof -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -9 ;;
    pmc) KWS_T3_STREAM=1 bash tools/stream_pmc.sh ;;
    timing) KWS_T3_STREAM=1 KWS_LIB=$PWD/honk2_amd/variants/lib_stream_t.so timeout -k 10 120 python tools/stream_debug.py big 2>&1 | grep "stream L" | sort | uniq -c | sort -rn | head -48 ;;
    ablate) for n in exp stream_a1 stream_a2 stream_a16 stream_a32; do
              lib=$PWD/honk2_amd/variants/lib_$n.so; [ $n = exp ] && lib=$PWD/honk2_amd/libkws_hip_exp.so
              echo -n "$n "; KWS_T3_STREAM=1 KWS_LIB=$lib bm | cut -c60-130
            done ;;
    *) echo "unknown: $what"; exit 2 ;;
  esac
done
