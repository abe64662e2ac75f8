#!/bin/bash
# round 4: B-fragment look-ahead depth of the pair / triple kernels (PAIR_BPF variants), res15 bf16 B = 4 096 and res26 fp16
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
for rep in 1 2; do
for v in libkws_hip.so variants/lib_bpf2.so variants/lib_bpf3.so; do
  echo "{\"lib\": \"$v\"}"
  KWS_LIB=$PWD/honk2_amd/$v KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c1-170
  KWS_LIB=$PWD/honk2_amd/$v KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res26 2>/dev/null | cut -c1-170
done
done | tee gpurun_out/r4/bpf_ab.txt
