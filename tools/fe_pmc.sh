#!/bin/bash
# counter passes over the front end (tools/fe_prof.py); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d gpurun_out/fe_pmc1 --output-format csv -- python3 tools/fe_prof.py > gpurun_out/fe_pmc1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d gpurun_out/fe_pmc2 --output-format csv -- python3 tools/fe_prof.py > gpurun_out/fe_pmc2.log 2>&1
