#!/bin/bash
# round 4: fused res8 iteration -- its parity tests, then new / variants alternating (R8_VARIANTS = names under honk2_amd/variants)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "res8 or fused or smoke or range or golden or reference" > gpurun_out/r4/r8_tests.txt 2>&1
rc=$?; tail -5 gpurun_out/r4/r8_tests.txt; [ $rc -eq 0 ] || exit $rc
{
for rep in 1 2; do
  R8_TAG="new" timeout -k 10 200 python tools/r8_time.py 2>/dev/null
  for v in $R8_VARIANTS; do KWS_LIB=$PWD/honk2_amd/variants/lib_$v.so R8_TAG="$v" timeout -k 10 200 python tools/r8_time.py 2>/dev/null; done
done
} | tee gpurun_out/r4/r8_iter.txt
