cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
V=$PWD/honk2_amd/variants
for rep in 1 2; do
  for n in exp fe_ms8 fe_ms16; do
    lib=$V/lib_$n.so; [ $n = exp ] && lib=$PWD/honk2_amd/libkws_hip_exp.so
    KWS_LIB=$lib FE_TAG=$n timeout -k 10 120 python tools/fe_time.py
  done
done
