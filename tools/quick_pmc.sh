#!/bin/bash
# Quick counters of one tools/bench_models.py configuration: tools/quick_pmc.sh <dtype> <batch> <model> [kernel substring]
dt=$1; bt=$2; model=$3; pat=${4:-conv3x3_tile}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/qpmc
rm -rf $o; mkdir -p $o
n=1
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  KWS_BENCH_DTYPE=$dt KWS_BENCH_BATCH=$bt timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs -d $o/pmc$n --output-format csv -- python3 tools/bench_models.py $model > $o.pmc$n.log 2>&1 || { echo "pmc $n failed"; tail -5 $o.pmc$n.log; exit 1; }
  n=$((n+1))
done
python3 - "$o" "$pat" <<'PY'
import sys, glob, csv, collections, os
src, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for path in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    ms = sum(dur[k]) / max(len(dur[k]), 1)
    out = {"kernel": k[-60:], "avg_ms(pmc runs)": round(ms, 4), "calls": len(dur[k])}
    if "FETCH_SIZE" in m: out["fetch_MB"] = round(m["FETCH_SIZE"] / 1e3 * 2, 1)     # KB; gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM)
    if "WRITE_SIZE" in m: out["write_MB"] = round(m["WRITE_SIZE"] / 1e3, 1)
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m: out["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128), 3)
    if "fetch_MB" in out and "write_MB" in out and ms: out["TBps"] = round((out["fetch_MB"] + out["write_MB"]) / ms / 1e3, 2)
    out["raw"] = {c: round(v) for c, v in m.items()}
    print(out)
PY
