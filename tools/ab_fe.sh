# A/B of front-end library variants: tools/ab_fe.sh <variant> [<variant> ...]  (honk2_amd/variants/lib_<name>.so; "default" = the built library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
V=$PWD/honk2_amd/variants
for rep in 1 2; do
  for n in default "$@"; do
    if [ $n = default ]; then FE_TAG=$n timeout -k 10 120 python tools/fe_time.py; else KWS_LIB=$V/lib_$n.so FE_TAG=$n timeout -k 10 120 python tools/fe_time.py; fi
  done
done
