#!/bin/bash
# (r4: the stats run carries --no-cpu-baseline too: the literal-tone parity record launches the kernels on ONE clip, and two 0.06 ms launches among 31 pull rocprof's average from 10.4 to 9.8 ms)
# Per-round profiles (run on the GPU box from the repo root: tools/profile_round.sh <tag> [round dir, default r04]): rocprofv3 kernel stats of the default bench command, PMC passes for
# its two kernels, and kernel stats + PMC of BASELINE configs[2] / [4] in their own dtype.  Summaries -> gpurun_out/prof_<tag>*; tools/summarize_prof.py <dir> <tag> profiles/<round> turns them into the committed files
tag=${1:-v9}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
rm -rf $out ${out}_res15_bf16 ${out}_cnn_fp16 ${out}_res15_f32   # (gpurun_out/ survives between rounds: never mix in an older run's files)
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-shard --no-h2d --no-secondary --no-cpu-baseline --no-live-traffic > $out.stats.log 2>&1 || { echo stats failed; tail -3 $out.stats.log; exit 1; }
n=1
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs -d $out/pmc$n --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-shard --no-h2d --no-live-traffic > $out.pmc$n.log 2>&1 || { echo "pmc $n failed"; tail -3 $out.pmc$n.log; exit 1; }
  n=$((n+1))
done
for cfg in "res15_bf16:bf16:4096:resnet__res15" "cnn_fp16:fp16:8192:cnn__cnn-trad-pool2" "res15_f32:f32:2048:resnet__res15"; do
  IFS=: read name dt bt model <<< "$cfg"
  o=gpurun_out/prof_${tag}_$name
  mkdir -p $o
  KWS_BENCH_DTYPE=$dt KWS_BENCH_BATCH=$bt timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $o/stats --output-format csv -- python3 tools/bench_models.py $model > $o.stats.log 2>&1 || { echo "$name stats failed"; exit 1; }
  n=1
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    KWS_BENCH_DTYPE=$dt KWS_BENCH_BATCH=$bt timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs -d $o/pmc$n --output-format csv -- python3 tools/bench_models.py $model > $o.pmc$n.log 2>&1 || { echo "$name pmc $n failed"; exit 1; }
    n=$((n+1))
  done
done
echo profiles done
