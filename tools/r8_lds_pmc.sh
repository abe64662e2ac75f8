#!/bin/bash
# LDS bank conflicts of the fused res8 kernel by phase (round 4): KWS_R8_DEBUG 0 = all, 1 = without conv_0's window reads / MFMAs, 2 = without the k-loops (epilogue stores, conv_0,
# staging and tail remain), 3 = without both (staging, epilogue stores, tail).  Results are wrong by construction with a bit set; only the counters matter.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r8lds; rm -rf $o; mkdir -p $o
for d in 0 1 2 3; do
  KWS_R8_DEBUG=$d R8_REPS=3 R8_SETTLE_S=0 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT -d $o/d$d --output-format csv -- python3 tools/r8_time.py > $o.d$d.log 2>&1 || { echo "debug $d failed"; tail -5 $o.d$d.log; }
done
python3 - $o <<'PY'
import sys, glob, csv, collections, os
for d in range(4):
    agg = collections.defaultdict(list)
    for path in glob.glob(os.path.join(sys.argv[1], f"d{d}", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "res8h_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {c: sum(v) / len(v) for c, v in agg.items()}
    per = 65536 * 4
    print("KWS_R8_DEBUG", d, {c: round(v / per, 1) for c, v in sorted(m.items())}, "(per wave and clip)")
PY
