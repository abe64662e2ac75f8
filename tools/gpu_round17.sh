#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
V=$PWD/honk2_amd/variants
KWS_LIB=$V/lib_t3timing.so KWS_T3_TIMING=$PWD/gpurun_out/t3_ts.bin KWS_BENCH_BATCH=1024 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>&1 | tail -1 | cut -c1-100
python3 tools/t3_phases.py gpurun_out/t3_ts.bin | head -4
