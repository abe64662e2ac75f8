# conv_band k-loop ablations (results wrong): build the variants first -- for a in 1 2 3; do tools/variant.sh band_abl$a conv_band.hip "-DBAND_ABLATE=$a"; done
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for n in default band_abl1 band_abl2 band_abl3 default; do
  lib=$PWD/honk2_amd/variants/lib_$n.so; [ $n = default ] && lib=$PWD/honk2_amd/libkws_hip.so
  echo "{\"variant\": \"$n\"}"
  KWS_LIB=$lib KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 2>/dev/null | cut -c1-130
  KWS_LIB=$lib timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 2>/dev/null | cut -c1-130
done
