set -e
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/flags.jsonl; : > $out
for rep in 1 2; do
for n in default r8_ilp r8_mem r8_nopost r8_trk r8_iter r8_pad20 r8_wprio; do
  if [ $n = default ]; then R8_TAG=$n timeout -k 10 120 python tools/r8_time.py >> $out; else KWS_LIB=$PWD/honk2_amd/variants/lib_$n.so R8_TAG=$n timeout -k 10 120 python tools/r8_time.py >> $out; fi
done
for n in default fe_ilp fe_mem fe_nopost fe_trk; do
  if [ $n = default ]; then FE_TAG=$n timeout -k 10 120 python tools/fe_time.py >> $out; else KWS_LIB=$PWD/honk2_amd/variants/lib_$n.so FE_TAG=$n timeout -k 10 120 python tools/fe_time.py >> $out; fi
done
done
cat $out
