set -e
cd /root/repo
export TMPDIR=/tmp
V=$PWD/honk2_amd/variants
for rep in 1 2; do
FE_TAG=default timeout -k 10 120 python tools/fe_time.py
for n in fe_abl1 fe_abl2 fe_abl3; do KWS_LIB=$V/lib_$n.so FE_TAG=$n timeout -k 10 120 python tools/fe_time.py; done
done
