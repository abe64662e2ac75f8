set -e
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/flags4.jsonl; : > $out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/flags4_tests.log 2>&1 || { tail -30 gpurun_out/flags4_tests.log; exit 1; }
tail -2 gpurun_out/flags4_tests.log
V=$PWD/honk2_amd/variants
for rep in 1 2; do
  KWS_LIB=$V/lib_prev.so R8_TAG=prev timeout -k 10 120 python tools/r8_time.py >> $out
  R8_TAG=new timeout -k 10 120 python tools/r8_time.py >> $out
  KWS_LIB=$V/lib_prev.so FE_TAG=prev timeout -k 10 120 python tools/fe_time.py >> $out
  FE_TAG=new timeout -k 10 120 python tools/fe_time.py >> $out
done
for n in prev new; do
  lib=$V/lib_prev.so; [ $n = new ] && lib=$PWD/honk2_amd/libkws_hip.so
  echo "{\"variant\": \"$n\"}" >> $out
  KWS_LIB=$lib timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 cnn__cnn-trad-fpool3 cnn__cnn-one-fstride4 resnet__res15 resnet__res26 >> $out 2>/dev/null
  KWS_LIB=$lib KWS_BENCH_DTYPE=bf16x3 timeout -k 10 300 python tools/bench_models.py resnet__res15 >> $out 2>/dev/null
done
cut -c1-200 $out
