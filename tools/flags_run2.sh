set -e
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/flags2.jsonl; : > $out
V=$PWD/honk2_amd/variants
for rep in 1 2; do
for n in default r8_nopost r8np_ord1 r8np_bd2 r8np_ilp r8np_nomis r8_nomis; do
  if [ $n = default ]; then R8_TAG=$n timeout -k 10 120 python tools/r8_time.py >> $out; else KWS_LIB=$V/lib_$n.so R8_TAG=$n timeout -k 10 120 python tools/r8_time.py >> $out; fi
done
done
for n in default np_conv3x3_tile np_layerwise_bf16x6; do
  lib=$V/lib_$n.so; [ $n = default ] && lib=$PWD/honk2_amd/libkws_hip.so
  echo "{\"variant\": \"$n\"}" >> $out
  KWS_LIB=$lib KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res15 >> $out 2>/dev/null
  KWS_LIB=$lib KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=4096 timeout -k 10 300 python tools/bench_models.py resnet__res26 >> $out 2>/dev/null
  KWS_LIB=$lib timeout -k 10 300 python tools/bench_models.py resnet__res15 resnet__res8_narrow >> $out 2>/dev/null
done
for n in default np_conv_band np_conv_in1 np_layerwise_bf16x6; do
  lib=$V/lib_$n.so; [ $n = default ] && lib=$PWD/honk2_amd/libkws_hip.so
  echo "{\"variant\": \"$n\"}" >> $out
  KWS_LIB=$lib KWS_BENCH_DTYPE=fp16 KWS_BENCH_BATCH=8192 timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 >> $out 2>/dev/null
  KWS_LIB=$lib timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 cnn__cnn-trad-fpool3 cnn__cnn-tstride4 >> $out 2>/dev/null
done
cut -c1-200 $out
