#!/bin/bash
# rocprofv3 passes over `bench.py` (run on the GPU box from the repo root): kernel stats of the default command, then
# counter passes (FETCH_SIZE and WRITE_SIZE each need a pass of their own).
# usage: tools/profile_bench.sh <tag>   -> gpurun_out/prof_<tag>/{stats,pmc1..4}
tag=${1:-cur}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
mkdir -p $out
if [ -z "$PMC_ONLY" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 > $out.stats.log 2>&1 || exit 1
fi
n=1
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $ctrs -d $out/pmc$n --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out.pmc$n.log 2>&1 || exit 1
  n=$((n+1))
done
