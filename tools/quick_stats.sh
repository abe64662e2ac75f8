# per-kernel average durations of one tools/bench_models.py configuration: tools/quick_stats.sh [dtype] [batch] [model]   (env such as KWS_T3_DEBUG passes through)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
dt=${1:-bf16}; bt=${2:-4096}; model=${3:-resnet__res15}
o=gpurun_out/qstats; rm -rf $o; mkdir -p $o
KWS_BENCH_DTYPE=$dt KWS_BENCH_BATCH=$bt timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $o --output-format csv -- python3 tools/bench_models.py $model > $o.log 2>&1
python3 - <<'PY'
import csv, glob
for path in glob.glob('gpurun_out/qstats/**/*_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'kws::' in r['Name']:
            print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us', r['Percentage'])
PY
