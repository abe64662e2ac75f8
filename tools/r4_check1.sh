#!/bin/bash
# round 4, first GPU pass: the graph/memset probe, the new tests on their own, then the whole GPU suite and one bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout -k 10 120 tools/graph_memset_probe gpurun_out/r4/graph_probe.dot > gpurun_out/r4/graph_probe.txt 2>&1; echo "probe rc $?" >> gpurun_out/r4/graph_probe.txt
cat gpurun_out/r4/graph_probe.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "back_to_back or absurd or real_data_parallel or two_ranks or one_rank_group or graph_capturable" > gpurun_out/r4/new_tests.txt 2>&1
rc=$?; tail -15 gpurun_out/r4/new_tests.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4/all_tests.txt 2>&1
rc=$?; tail -5 gpurun_out/r4/all_tests.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r4/bench1.json 2> gpurun_out/r4/bench1.err
rc=$?; tail -c 600 gpurun_out/r4/bench1.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/bench1.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["frontend"]["kernel_ms"], d["roofline"]["kernel_ms"])
print(json.dumps(d.get("parity_literal_tone"), indent=1))
print(json.dumps(d["cpu_baseline"], indent=1)[:2500])
PY
exit $rc
