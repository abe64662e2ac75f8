#!/bin/bash
# the default bench command as the driver runs it, timed by the wall clock: python bench.py  (the line carries roofline.traffic measured live)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
t0=$(date +%s)
timeout -k 10 600 python bench.py > gpurun_out/live_bench.json 2> gpurun_out/live_bench.err; rc=$?
t1=$(date +%s)
echo "rc $rc wall $((t1 - t0)) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/live_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print({k: r.get(k) for k in ("frac","kernel_ms","traffic","traffic_static_committed","traffic_live")}); print(r.get("traffic_source"))
print(d["value"], d["ms_per_step"])
PY
