#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for ch in 1024 256 128 96 64 48 32; do
  echo "chunk $ch"
  KWS_TILED_CHUNK=$ch KWS_BENCH_BATCH=2048 timeout -k 10 300 python tools/bench_models.py resnet__res15 resnet__res26 2>/dev/null | cut -c1-200 || exit 1
done
