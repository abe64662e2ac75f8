cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for st in 1 0; do for dt in fp16 bf16; do echo -n "stream=$st $dt "; KWS_T3_STREAM=$st KWS_BENCH_DTYPE=$dt KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res26 2>/dev/null | cut -c60-140; done; done
for st in 1 0; do echo -n "stream=$st res15 fp16 "; KWS_T3_STREAM=$st KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c60-140; done
