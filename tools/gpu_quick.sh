#!/bin/bash
# quick checkpoint: GPU suite, smoke, bench.  tools/gpu_quick.sh <tag> [pytest -k expr]   (files: gpurun_out/<tag>_*)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=${1:-q}
if [ -n "$2" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "$2" > gpurun_out/${tag}_tests.log 2>&1
else
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/${tag}_tests.log 2>&1
fi
rc=$?; tail -3 gpurun_out/${tag}_tests.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; grep -n "Error\|assert\|FAILED" gpurun_out/${tag}_tests.log | head -20; exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.log 2> gpurun_out/${tag}_bench.err || { echo bench failed; tail -5 gpurun_out/${tag}_bench.err; exit 1; }
grep "^{" gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench.json
python3 - gpurun_out/${tag}_bench.json <<'PY'
import json
import sys
d=json.load(open(sys.argv[1]))
print('bench', round(d['value']), round(d['ms_per_step'],3), 'r8', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4), 'fe', round(d['frontend']['kernel_ms'],3), d['parity']['max_abs_err'], d['parity']['argmax_equal'], 'cpu', round(d['cpu_baseline']['value']))
for r in d.get('secondary', []):
    print(' ', r['config'], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (r.get('features_to_logits') or r.get('wav_to_logits')).items()}, r.get('efficiency_vs_full_batch'), r['parity']['pass'])
print('  shard', d['shard']['efficiency_vs_full_batch'])
PY
