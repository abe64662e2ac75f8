# kernel time against batch size (fixed cost per launch = start-up + tail): tools/batch_sweep.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for b in 1024 2048 4096 8192 16384 32768 65536; do FE_B=$b FE_TAG=B$b timeout -k 10 120 python tools/fe_time.py | cut -c1-70; done
for b in 1024 2048 4096 8192 16384 32768 65536; do R8_B=$b R8_TAG=B$b R8_REPS=20 timeout -k 10 120 python tools/r8_time.py | cut -c1-110; done
