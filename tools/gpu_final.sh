#!/bin/bash
# end-of-round checkpoint: full GPU suite, smoke, bench, per-model table, profiles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r2_tests_final.log 2>&1
rc=$?; tail -3 gpurun_out/r2_tests_final.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; grep -n "Error\|assert" gpurun_out/r2_tests_final.log | head; exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_final.log 2>&1 || { echo bench failed; tail -5 gpurun_out/r2_bench_final.log; exit 1; }
grep "^{" gpurun_out/r2_bench_final.log > gpurun_out/r2_bench_final.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench_final.json'))
print('bench', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4), round(d['frontend']['kernel_ms'],3), d['parity']['max_abs_err'], d['parity']['argmax_equal'], round(d['cpu_baseline']['value']))
PY
timeout -k 10 600 python tools/bench_models.py > gpurun_out/r2_models_final.jsonl 2>/dev/null || exit 1
KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 >> gpurun_out/r2_models_final.jsonl 2>/dev/null || exit 1
KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 >> gpurun_out/r2_models_final.jsonl 2>/dev/null || exit 1
cut -c1-175 gpurun_out/r2_models_final.jsonl
bash tools/profile_r02.sh final
