set -e
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/flags3.jsonl; : > $out
V=$PWD/honk2_amd/variants
for rep in 1 2; do
for n in default r8_nopost r8_fence r8np_ord2; do
  if [ $n = default ]; then R8_TAG=$n timeout -k 10 120 python tools/r8_time.py >> $out; else KWS_LIB=$V/lib_$n.so R8_TAG=$n timeout -k 10 120 python tools/r8_time.py >> $out; fi
done
for n in default fe_trk fe_ord1 fenp_ord1 fetrk_ord1; do
  if [ $n = default ]; then FE_TAG=$n timeout -k 10 120 python tools/fe_time.py >> $out; else KWS_LIB=$V/lib_$n.so FE_TAG=$n timeout -k 10 120 python tools/fe_time.py >> $out; fi
done
done
for n in default band_ord1 npband_ord1; do
  lib=$V/lib_$n.so; [ $n = default ] && lib=$PWD/honk2_amd/libkws_hip.so
  echo "{\"variant\": \"$n\"}" >> $out
  KWS_LIB=$lib timeout -k 10 300 python tools/bench_models.py cnn__cnn-trad-pool2 cnn__cnn-trad-fpool3 cnn__cnn-tstride4 cnn__cnn-tpool2 >> $out 2>/dev/null
done
cut -c1-200 $out
