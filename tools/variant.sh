#!/bin/bash
# Build an A/B variant of the library: tools/variant.sh <name> <file.hip> "<extra hipcc flags>"  -> honk2_amd/variants/lib_<name>.so
# (only <file.hip> is recompiled with the extra flags; the other objects come from the EXPERIMENTS=1 build, so every variant carries the debug / timing switches)
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../honk2_amd/csrc"
make -s -j8 EXPERIMENTS=1 >/dev/null
mkdir -p ../variants build_exp/var_$name
case "$src" in frontend_f16x3.hip) flags="$flags -fno-slp-vectorize -mllvm -amdgpu-use-amdgpu-trackers";; res8_f16x3.hip) flags="$flags -fno-slp-vectorize";; conv_band.hip) flags="$flags -mllvm -disable-post-ra";; conv3x3_tile.hip) flags="$flags -mllvm -amdgpu-sched-strategy=max-ilp";; esac   # the Makefile's per-file flag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc -DKWS_EXPERIMENTS $flags -x hip -c ${VARIANT_SRC:-$src} -o build_exp/var_$name/$src.o
objs=""
for f in kws_api.cpp frontend.hip frontend_f16x3.hip res8_fused.hip res8_bf16x6.hip res8_f16x3.hip layerwise.hip layerwise_bf16x6.hip conv3x3_tile.hip conv3x3_stream.hip conv_band.hip conv_cols.hip conv_in1.hip; do
  if [ "$f" = "$src" ]; then objs="$objs build_exp/var_$name/$f.o"; else objs="$objs build_exp/$f.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o ../variants/lib_$name.so
echo built honk2_amd/variants/lib_$name.so
