#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
V=$PWD/honk2_amd/variants
for lib in default t3_st1 t3_st2 t3_st4 default; do
  if [ $lib = default ]; then unset KWS_LIB; else export KWS_LIB=$V/lib_$lib.so; fi
  echo "== $lib"
  timeout -k 10 300 python tools/bench_models.py resnet__res15 resnet__res26 2>/dev/null | cut -c1-140 || exit 1
done
