#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
V=$PWD/honk2_amd/variants
for rep in 1 2; do
echo "== default"; timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "model_logits and cnn" 2>&1 | tail -3
echo "== old generic kernel"; KWS_LIB=$V/lib_lwold.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "model_logits and cnn" 2>&1 | tail -3
done
echo "== default, bf16 parts"; KWS_MATRIX_PARTS=bf16 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "model_logits and cnn" 2>&1 | tail -3
