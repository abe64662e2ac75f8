#!/bin/bash
# checkpoint: full GPU suite, smoke, bench, per-model table, profiles.  tools/gpu_checkpoint.sh <tag> (files: gpurun_out/<tag>_*)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=${1:-final}
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/${tag}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${tag}_tests.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; grep -n "Error\|assert" gpurun_out/${tag}_tests.log | head; exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.log 2>&1 || { echo bench failed; tail -5 gpurun_out/${tag}_bench.log; exit 1; }
grep "^{" gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench.json
python3 - gpurun_out/${tag}_bench.json <<'PY'
import json
import sys
d=json.load(open(sys.argv[1]))
print('bench', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4), round(d['frontend']['kernel_ms'],3), d['parity']['max_abs_err'], d['parity']['argmax_equal'], round(d['cpu_baseline']['value']))
PY
timeout -k 10 600 python tools/bench_models.py > gpurun_out/${tag}_models.jsonl 2>/dev/null || exit 1
KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 >> gpurun_out/${tag}_models.jsonl 2>/dev/null || exit 1
KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 >> gpurun_out/${tag}_models.jsonl 2>/dev/null || exit 1
cut -c1-175 gpurun_out/${tag}_models.jsonl
bash tools/profile_round.sh $tag r05
