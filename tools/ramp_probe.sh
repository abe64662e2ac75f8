# does a short measurement under-state the shard's rate?  the same 8 192-clip launches timed over 10, 50, 250 and 1000 repetitions
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export FE_SETTLE_S=0 R8_SETTLE_S=0     # from idle: that is the point
for r in 10 50 250 1000; do FE_B=8192 FE_REPS=$r FE_TAG=reps$r timeout -k 10 120 python tools/fe_time.py | cut -c1-70; done
for r in 10 50 250 1000; do R8_B=8192 R8_REPS=$r R8_TAG=reps$r timeout -k 10 120 python tools/r8_time.py | cut -c1-110; done
FE_B=65536 FE_REPS=30 FE_TAG=full timeout -k 10 120 python tools/fe_time.py | cut -c1-70
R8_B=65536 R8_REPS=30 R8_TAG=full timeout -k 10 120 python tools/r8_time.py | cut -c1-110
