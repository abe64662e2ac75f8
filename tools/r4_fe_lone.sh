#!/bin/bash
# round 4: the front end with one workgroup per CU against two -- launch time and phase stamps (how latency-bound are the phases?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
for n in 2 1; do
  KWS_FE_WGS_PER_CU=$n FE_TAG="wgs=$n" timeout -k 10 200 python tools/fe_time.py 2>/dev/null
  KWS_LIB=$PWD/honk2_amd/variants/lib_fets.so KWS_FE_WGS_PER_CU=$n timeout -k 10 200 python tools/fe_phases.py 2>/dev/null
done | tee gpurun_out/r4/fe_lone.txt
