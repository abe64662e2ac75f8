cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/qstats; rm -rf $o; mkdir -p $o
KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $o --output-format csv -- python3 tools/bench_models.py resnet__res15 > $o.log 2>&1
python3 - <<'PY'
import csv, glob
for path in glob.glob('gpurun_out/qstats/**/*_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'kws::' in r['Name']:
            print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us', r['Percentage'])
PY
