cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
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
bash tools/cnn_streams.sh trace >> $o 2>&1
echo "== per-kernel averages, one stream (rocprofv3 --kernel-trace --stats): product library, then -DCOLS_ABLATE variants (1 no weight loads, 2 no LDS fragment reads, 4 no stores, 8 no image DMA)" >> $o
STREAMS=0 VARIANTS="prod cols_a1 cols_a2 cols_a4 cols_a8" bash tools/cnn_streams.sh kstats >> $o 2>&1
echo "== conv_cols_kernel phases per unit (-DCOLS_TIMING, tools/cols_phases.py; 1 024 clips): k-loop / barrier / epilogue / image wait; two CUs, their two workgroups' units [start, k-loop end, barrier, epilogue end, image landed]" >> $o
bash tools/cnn_streams.sh phases >> $o 2>&1
echo "== counters (one rocprofv3 --pmc pass each, 1 024 clips, one stream)" >> $o
bash tools/cnn_streams.sh pmc >> $o 2>&1
tail -5 $o
