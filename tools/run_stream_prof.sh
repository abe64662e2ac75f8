cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/sprof
KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/sprof --output-format csv -- python3 tools/bench_models.py resnet__res15 > gpurun_out/sprof.log 2>&1
f=$(find gpurun_out/sprof -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -12
