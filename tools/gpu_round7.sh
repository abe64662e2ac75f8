#!/bin/bash
# PMC passes over the fused res8 kernel alone (tools/r8_time.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_r8b
mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1
n=1
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
            "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_IFETCH" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  R8_REPS=3 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs -d $out/pmc$n --output-format csv -- python3 tools/r8_time.py > $out.pmc$n.log 2>&1 || { echo "pass $n failed"; tail -5 $out.pmc$n.log; }
  n=$((n+1))
done
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list)
for path in glob.glob('gpurun_out/prof_r8b/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'res8h_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k, sum(v)/len(v), len(v))
PY
