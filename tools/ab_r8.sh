# A/B of fused-res8 library variants: tools/ab_r8.sh <variant> [<variant> ...]  (names under honk2_amd/variants/lib_<name>.so; "default" = the built library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
V=$PWD/honk2_amd/variants
for rep in 1 2; do
  for n in default "$@"; do
    if [ $n = default ]; then R8_TAG=$n timeout -k 10 120 python tools/r8_time.py; else KWS_LIB=$V/lib_$n.so R8_TAG=$n timeout -k 10 120 python tools/r8_time.py; fi
  done
done
