#!/bin/bash
# ONE A/B driver for library variants (built by tools/variant.sh into honk2_amd/variants/lib_<name>.so; "exp" = the EXPERIMENTS=1 build, "prod" = the
# product library).  Run on the GPU box from the repo root:   tools/ab.sh <what> <variant> [<variant> ...]       (each variant twice, alternating)
#   what = r8      fused res8 kernel alone, 65 536 clips                (tools/r8_time.py)
#          fe      front-end kernel alone, 65 536 clips                  (tools/fe_time.py)
#          power   fused res8 with rocm-smi clock / package power        (tools/r8_power.py)
#          models  tools/bench_models.py; models and dtype / batch from AB_MODELS, KWS_BENCH_DTYPE, KWS_BENCH_BATCH
# (replaces the per-kernel ab_*.sh / r4_*.sh / band_abl.sh drivers of rounds 3 - 4)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
what=$1; shift
for rep in 1 2; do
  for n in "$@"; do
    case $n in exp) lib=$PWD/honk2_amd/libkws_hip_exp.so;; prod) lib=$PWD/honk2_amd/libkws_hip.so;; *) lib=$PWD/honk2_amd/variants/lib_$n.so;; esac
    case $what in
      r8) KWS_LIB=$lib R8_TAG=$n timeout -k 10 120 python tools/r8_time.py 2>/dev/null ;;
      fe) KWS_LIB=$lib FE_TAG=$n timeout -k 10 120 python tools/fe_time.py 2>/dev/null ;;
      power) KWS_LIB=$lib timeout -k 10 120 python tools/r8_power.py 2>/dev/null ;;
      models) echo -n "$n "; KWS_LIB=$lib timeout -k 10 300 python tools/bench_models.py $AB_MODELS 2>/dev/null | cut -c1-170 ;;
      *) echo "unknown: $what"; exit 2 ;;
    esac
  done
done
