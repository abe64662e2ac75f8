#!/bin/bash
# cnn plans (run_cnn: two streams; conv_cols.hip): tests, A/B runs, chunk sizes, kernel timeline, per-kernel averages of library variants, phase stamps, counters.
#   tools/cnn_streams.sh [test] [ab] [cols] [ablate] [kstats] [phases] [pmc] [chunks] [stats] [trace] [evidence]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
for what in "$@"; do case $what in
test) timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout 600 -k "cnn or range_guard or chunk_loops or capturable or neighbours or two_stream or range_free" > gpurun_out/cs_tests.log 2>&1; tail -3 gpurun_out/cs_tests.log ;;
ab) for rep in 1 2; do for st in 0 1; do echo "streams=$st"; KWS_CNN_STREAMS=$st KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 200 python tools/bench_models.py cnn__cnn-trad-pool2 cnn__cnn-one-fstride4 cnn__cnn-tstride4 2>/dev/null | cut -c1-150; done; done ;;
cols) for rep in 1 2; do for st in 0 1; do echo "cols=$st"; KWS_CNN_COLS=$st KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 200 python tools/bench_models.py cnn__cnn-trad-pool2 cnn__cnn-trad-fpool3 2>/dev/null | cut -c1-150; done; done ;;
ablate) for rep in 1 2; do for v in ${VARIANTS:-cols_a0 cols_a1 cols_a2 cols_a3 cols_a4 cols_a8 cols_a16}; do echo -n "$v "; KWS_LIB=$PWD/honk2_amd/variants/lib_$v.so KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 200 python tools/bench_models.py cnn__cnn-trad-pool2 2>/dev/null | cut -c50-120; done; done ;;
kstats) for v in ${VARIANTS:-prod}; do case $v in prod) lib=$PWD/honk2_amd/libkws_hip.so;; exp) lib=$PWD/honk2_amd/libkws_hip_exp.so;; *) lib=$PWD/honk2_amd/variants/lib_$v.so;; esac
    rm -rf gpurun_out/cs_k_$v; KWS_LIB=$lib KWS_CNN_STREAMS=${STREAMS:-1} KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cs_k_$v -o t -- python3 tools/bench_models.py ${MODEL:-cnn__cnn-trad-pool2} > gpurun_out/cs_k_$v.log 2>&1
    echo "$v: $(grep -h 'conv_cols\|conv_band_kernel\|conv_in1' gpurun_out/cs_k_$v/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110 | tr '
' ' ')"; done ;;
pmc) o=gpurun_out/cs_pmc; rm -rf $o; mkdir -p $o; n=1
   for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM"; do
     KWS_CNN_STREAMS=0 KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=1024 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs -d $o/pmc$n --output-format csv -- python3 tools/bench_models.py cnn__cnn-trad-pool2 > $o.pmc$n.log 2>&1 || { echo "pmc $n failed"; tail -5 $o.pmc$n.log; }
     n=$((n+1)); done
   python3 - $o <<'PY'
import sys, glob, csv, collections, os
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "conv_cols" in r["Kernel_Name"] or "conv_band_kernel" in r["Kernel_Name"] or "conv_in1" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k, {c: round(v / 1e6, 2) for c, v in m.items()})
    print("   of wave cycles:", {c: round(m[c] / wc, 3) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_VMEM") if c in m})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m: print("   mfma busy:", round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8), 3))
PY
   ;;
phases) KWS_LIB=$PWD/honk2_amd/variants/lib_${VARIANT:-cols_t}.so KWS_CNN_STREAMS=0 KWS_BAND_TIMING=$PWD/gpurun_out/cols_ts.bin KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=1024 timeout -k 10 200 python tools/bench_models.py cnn__cnn-trad-pool2 > /dev/null 2>&1; python3 tools/cols_phases.py gpurun_out/cols_ts.bin ;;
evidence)   # everything profiles/r05/cnn_cols_evidence.txt holds, in one run (variants cols_t, cols_a1 / a2 / a4 / a8 from tools/variant.sh first)
o=gpurun_out/cnn_cols_evidence.txt
{
echo "== cnn-trad-pool2 fp16, B = 8192 (tools/bench_models.py): one stream / two streams x conv_band / conv_cols, two alternating rounds"
for rep in 1 2; do for st in 0 1; do for c in 0 1; do echo -n "streams=$st cols=$c  "; KWS_CNN_STREAMS=$st KWS_CNN_COLS=$c KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 200 python tools/bench_models.py cnn__cnn-trad-pool2 2>/dev/null | cut -c55-200; done; done; done
echo "== other cnn models, fp16 and f32, streams 0 / 1"
for st in 0 1; do echo "streams=$st"; KWS_CNN_STREAMS=$st KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-fpool3 cnn__cnn-tstride4 cnn__cnn-tpool2 cnn__cnn-one-fstride4 2>/dev/null | cut -c1-130; KWS_CNN_STREAMS=$st timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 cnn__cnn-tstride8 cnn__cnn-one-fstride8 2>/dev/null | cut -c1-130; done
echo "== power / clock while cnn-trad-pool2 fp16 loops (rocm-smi)"
for c in 0 1; do echo -n "cols=$c "; KWS_CNN_COLS=$c KWS_BENCH_POWER=1 KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 python tools/bench_models.py cnn__cnn-trad-pool2 2>/dev/null | cut -c55-400; done
} > $o 2>&1
echo "== kernel timeline of one call (rocprofv3 --kernel-trace; start, end, duration in us; q1 = caller's stream, q2 = the handle's)" >> $o
bash "$0" trace >> $o 2>&1
echo "== per-kernel averages, one stream (rocprofv3 --kernel-trace --stats): product library, then -DCOLS_ABLATE variants (1 no weight loads, 2 no LDS fragment reads, 4 no stores, 8 no image DMA)" >> $o
STREAMS=0 VARIANTS="prod cols_a1 cols_a2 cols_a4 cols_a8" bash "$0" kstats >> $o 2>&1
echo "== conv_cols_kernel phases per unit (-DCOLS_TIMING, tools/cols_phases.py; 1 024 clips): k-loop / barrier / epilogue / image wait; two CUs, their two workgroups' units [start, k-loop end, barrier, epilogue end, image landed]" >> $o
bash "$0" phases >> $o 2>&1
echo "== counters (one rocprofv3 --pmc pass each, 1 024 clips, one stream)" >> $o
bash "$0" pmc >> $o 2>&1
tail -5 $o
   ;;
chunks) for c in ${CHUNKS:-768 1024 1280 1536 2048 3072}; do echo "chunk=$c"; KWS_LIB=$PWD/honk2_amd/libkws_hip_exp.so KWS_CNN_CHUNK=$c KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=${BATCH:-12288} timeout -k 10 200 python tools/bench_models.py cnn__cnn-trad-pool2 cnn__cnn-tstride4 2>/dev/null | cut -c1-150; done ;;
stats) for c in ${CHUNKS:-1024 1536 2048}; do rm -rf gpurun_out/cs_stats_$c; KWS_LIB=$PWD/honk2_amd/libkws_hip_exp.so KWS_CNN_CHUNK=$c KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=${BATCH:-12288} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cs_stats_$c -o t -- python3 tools/bench_models.py ${MODEL:-cnn__cnn-trad-pool2} > gpurun_out/cs_stats_$c.log 2>&1; echo "chunk=$c"; cut -d, -f1-4 gpurun_out/cs_stats_$c/t_kernel_stats.csv | cut -c1-150; done ;;
trace) rm -rf gpurun_out/cs_trace; KWS_LIB=$PWD/honk2_amd/libkws_hip_exp.so KWS_CNN_CHUNK=${CHUNK:-0} KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=${BATCH:-8192} timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/cs_trace -o t -- python3 tools/bench_models.py cnn__cnn-trad-pool2 > gpurun_out/cs_trace.log 2>&1; python3 tools/trace_timeline.py gpurun_out/cs_trace/t_results.db 72 60 > gpurun_out/cs_timeline.txt; tail -60 gpurun_out/cs_timeline.txt ;;
esac; done
