#!/bin/bash
# round 4: front-end iteration -- its parity tests, launch time (two settled runs), phase stamps of a -DFE16_TIMING build when there is one
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "frontend or mfcc or pcm16 or windows or wav" > gpurun_out/r4/fe_tests.txt 2>&1
rc=$?; tail -5 gpurun_out/r4/fe_tests.txt; [ $rc -eq 0 ] || exit $rc
{
for rep in 1 2; do FE_TAG="new" timeout -k 10 200 python tools/fe_time.py 2>/dev/null; done
for v in $FE_VARIANTS; do KWS_LIB=$PWD/honk2_amd/variants/lib_$v.so FE_TAG="$v" timeout -k 10 200 python tools/fe_time.py 2>/dev/null; done
[ -f honk2_amd/variants/lib_fets.so ] && KWS_LIB=$PWD/honk2_amd/variants/lib_fets.so timeout -k 10 200 python tools/fe_phases.py 2>/dev/null
} | tee gpurun_out/r4/fe_iter.txt
