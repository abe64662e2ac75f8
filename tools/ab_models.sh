# A/B of the per-model table rows: tools/ab_models.sh <variant> -- models...   (KWS_BENCH_DTYPE / KWS_BENCH_BATCH from the environment)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
v=$1; shift
for n in prev new prev new; do
  lib=$PWD/honk2_amd/variants/lib_$v.so; [ $n = new ] && lib=$PWD/honk2_amd/libkws_hip.so
  echo "{\"variant\": \"$n\"}"
  KWS_LIB=$lib timeout -k 10 300 python tools/bench_models.py "$@" 2>/dev/null | cut -c1-150
done
