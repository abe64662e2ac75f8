#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
V=$PWD/honk2_amd/variants
for lib in default t3_192 t3_448 default t3_192; do
  if [ $lib = default ]; then unset KWS_LIB; else export KWS_LIB=$V/lib_$lib.so; fi
  echo "== $lib"
  timeout -k 10 300 python tools/bench_models.py resnet__res15 resnet__res26 resnet__res26_narrow 2>/dev/null | cut -c1-200 || exit 1
done
