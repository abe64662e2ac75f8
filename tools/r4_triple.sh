#!/bin/bash
# round 4: the three-layer kernel -- bit-identity tests, then res15 bf16 (configs[2]) at B = 4 096 with triples on / off and the tile configurations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "triples or layer_pairs" > gpurun_out/r4/triple_tests.txt 2>&1
rc=$?; tail -15 gpurun_out/r4/triple_tests.txt; [ $rc -eq 0 ] || exit $rc
for v in "KWS_T3_TRIPLE=0" "KWS_T3_TRIPLE=1" "KWS_T3_TRIPLE=1 KWS_T3_TRIPLE_CFG=2" "KWS_T3_TRIPLE=1 KWS_T3_TRIPLE_CFG=3" "KWS_T3_TRIPLE=0" "KWS_T3_TRIPLE=1"; do
  echo "{\"variant\": \"$v\"}"
  env $v KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c1-170
done | tee gpurun_out/r4/triple_ab.txt
for v in "KWS_T3_TRIPLE=0" "KWS_T3_TRIPLE=2"; do
  echo "{\"variant\": \"res26 fp16 $v\"}"
  env $v KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res26 2>/dev/null | cut -c1-170
done | tee -a gpurun_out/r4/triple_ab.txt
bash tools/quick_stats.sh bf16 4096 resnet__res15 | tee gpurun_out/r4/triple_kernels.txt
