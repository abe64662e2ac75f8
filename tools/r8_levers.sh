#!/bin/bash
# (r5) What fewer MFMAs would buy the fused res8 kernel AT THE POWER CAP, measured: the shipping kernel (EXPERIMENTS build) against variants that drop
# k-steps / position tile 20 (honk2_amd/variants/lib_r8_{k13,k12,nox,k13nox}.so, built by tools/variant.sh with -DR8H_ABLATE_K / -DR8H_ABLATE_NOX;
# results wrong by construction).  Per variant: ms per 65 536 clips, sclk and package power sampled from rocm-smi while 250 launches run, uJ per clip;
# then SQ_INSTS_MFMA per launch from one rocprofv3 --pmc pass.  Output: gpurun_out/r8_levers.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r8_levers.txt
: > $out
V=$PWD/honk2_amd/variants
E=$PWD/honk2_amd/libkws_hip_exp.so
for rep in 1 2; do
  for n in exp r8_k13 r8_nox r8_k13nox r8_k12 exp; do
    lib=$V/lib_$n.so; [ $n = exp ] && lib=$E
    KWS_LIB=$lib timeout -k 10 120 python tools/r8_power.py >> $out 2>/dev/null || { echo "power run of $n failed"; exit 1; }
  done
done
KWS_LIB=$E R8_ZERO=1 timeout -k 10 120 python tools/r8_power.py >> $out 2>/dev/null
echo "--- SQ_INSTS_MFMA per launch (65 536 clips)" >> $out
for n in exp r8_k13 r8_nox r8_k13nox r8_k12; do
  lib=$V/lib_$n.so; [ $n = exp ] && lib=$E
  d=gpurun_out/lev_pmc_$n; rm -rf $d
  KWS_LIB=$lib R8_REPS=2 R8_SETTLE_S=0 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $d --output-format csv -- python3 tools/r8_time.py > $d.log 2>&1 || { echo "pmc run of $n failed"; tail -3 $d.log; exit 1; }
  python3 - $d $n >> $out <<'PY'
import csv, glob, os, sys, collections
agg = collections.defaultdict(list)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "res8h_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print(sys.argv[2], {k: round(v) for k, v in m.items()}, "mfma_busy", round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128), 3))
PY
  rm -rf $d
done
cat $out
