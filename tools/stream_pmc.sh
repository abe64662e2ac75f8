#!/bin/bash
# wave-time breakdown of the stream kernels (res15 bf16, 1 024 clips): SQ wait / active counters (quad-cycles), one pass
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/spmc; rm -rf $o; mkdir -p $o
n=1
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=1024 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs -d $o/pmc$n --output-format csv -- python3 tools/bench_models.py resnet__res15 > $o.pmc$n.log 2>&1 || { echo "pmc $n failed"; tail -5 $o.pmc$n.log; exit 1; }
  n=$((n+1))
done
python3 - $o <<'PY'
import sys, glob, csv, collections, os
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "conv3x3_stream" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k, {c: round(v / 1e6, 1) for c, v in m.items()})
    print("   of wave cycles:", {c: round(m[c] / wc, 3) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if c in m})
PY
