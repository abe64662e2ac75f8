#!/bin/bash
# checkpoint: full GPU suite, bench, per-model table (f32 + configs[2]/[4] dtypes), rocprof kernel stats of configs[2]/[4]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r2_tests_10.log 2>&1
rc=$?; tail -3 gpurun_out/r2_tests_10.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r2_bench_10.log 2>&1 || { echo bench failed; tail -5 gpurun_out/r2_bench_10.log; exit 1; }
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r2_bench_10.log') if l.startswith('{')][-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['frontend']['kernel_ms'], d['parity'])
PY
timeout -k 10 600 python tools/bench_models.py > gpurun_out/r2_models_f32.jsonl 2>gpurun_out/r2_models.err || { tail -3 gpurun_out/r2_models.err; exit 1; }
KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 >> gpurun_out/r2_models_cfg.jsonl 2>>gpurun_out/r2_models.err || exit 1
KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 >> gpurun_out/r2_models_cfg.jsonl 2>>gpurun_out/r2_models.err || exit 1
KWS_BENCH_DTYPE=f32 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 >> gpurun_out/r2_models_cfg.jsonl 2>>gpurun_out/r2_models.err || exit 1
cat gpurun_out/r2_models_f32.jsonl gpurun_out/r2_models_cfg.jsonl
KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_res15_bf16 --output-format csv -- python3 tools/bench_models.py resnet__res15 > gpurun_out/prof_res15_bf16.log 2>&1 || { echo prof1 failed; tail -3 gpurun_out/prof_res15_bf16.log; }
KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cnn_fp16 --output-format csv -- python3 tools/bench_models.py cnn__cnn-trad-pool2 > gpurun_out/prof_cnn_fp16.log 2>&1 || { echo prof2 failed; tail -3 gpurun_out/prof_cnn_fp16.log; }
python3 tools/kstats.py gpurun_out/prof_res15_bf16; python3 tools/kstats.py gpurun_out/prof_cnn_fp16
