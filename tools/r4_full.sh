#!/bin/bash
# whole GPU suite + one bench line (checkpoint of the tree)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests -m gpu -x -q -s -k "hundred_back_to_back" > gpurun_out/r4/replay100.txt 2>&1
rc=$?; tail -4 gpurun_out/r4/replay100.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4/all_tests.txt 2>&1
rc=$?; tail -5 gpurun_out/r4/all_tests.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r4/bench2.json 2> gpurun_out/r4/bench2.err
rc=$?; tail -c 300 gpurun_out/r4/bench2.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/bench2.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["frontend"]["kernel_ms"], d["roofline"]["kernel_ms"])
for s in d.get("secondary", []): print(s.get("workload", s.get("config")), {k: (v.get("ms") if isinstance(v, dict) else v) for k, v in s.items() if k in ("features_to_logits", "wav_to_logits")})
PY
exit $rc
