#!/bin/bash
# cache-path counters of the fused res8 kernel (weight fragments: how many of the L1 requests go on to L2): tools/r8_cache_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r8cache; rm -rf $o; mkdir -p $o
n=1
for ctrs in "TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum"; do
  R8_REPS=3 R8_SETTLE_S=0 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs -d $o/pmc$n --output-format csv -- python3 tools/r8_time.py > $o.pmc$n.log 2>&1 || { echo "pmc $n failed"; tail -5 $o.pmc$n.log; }
  n=$((n+1))
done
python3 - $o <<'PY'
import sys, glob, csv, collections, os
agg = collections.defaultdict(list)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "res8h_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(c, round(sum(v) / len(v)), "per launch over", len(v), "launches")
PY
