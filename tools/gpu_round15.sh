#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for dbg in 0 16 1 17 0; do
  echo "== KWS_T3_DEBUG=$dbg"
  KWS_T3_DEBUG=$dbg timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c1-140 || exit 1
done
