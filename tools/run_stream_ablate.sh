cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for n in exp stream_a16 stream_a32 stream_a48 stream_a4; do
  lib=$PWD/honk2_amd/variants/lib_$n.so; [ $n = exp ] && lib=$PWD/honk2_amd/libkws_hip_exp.so
  echo -n "$n "; KWS_LIB=$lib KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 2>/dev/null | cut -c60-130
done
