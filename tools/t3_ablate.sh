# phase ablations of the tiled / pair 3x3 kernels on res15 bf16 (results wrong by construction): KWS_T3_DEBUG bits
#   1 no k-loop (tile) / no conv_i k-loop (pair)   64 no conv_{i+1} k-loop (pair)   2 no staging loads   4 no output stores   8 no residual loads (tile)   16 no weight-fragment loads after the first (tile)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for d in 0 2 4 8 16 1 65 6 71 0; do
  echo "{\"KWS_T3_DEBUG\": $d}"
  KWS_T3_DEBUG=$d KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c1-120
done
