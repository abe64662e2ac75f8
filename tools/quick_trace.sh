#!/bin/bash
# Per-kernel times of one tools/bench_models.py configuration: tools/quick_trace.sh <dtype> <batch> <model>
dt=$1; bt=$2; model=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/qtrace
rm -rf $o; mkdir -p $o
KWS_BENCH_DTYPE=$dt KWS_BENCH_BATCH=$bt timeout -k 10 200 rocprofv3 --kernel-trace -d $o --output-format csv -- python3 tools/bench_models.py $model > $o.log 2>&1 || { tail -5 $o.log; exit 1; }
grep '"model"' $o.log
python3 - "$o" <<'PY'
import csv, glob, collections, os, sys
d = collections.defaultdict(list)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")[-64:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{k:66s} calls {len(v):5d} avg {sum(v)/len(v):8.4f} ms  share {sum(v)/tot:5.1%}")
PY
