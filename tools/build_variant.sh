#!/bin/bash
# Experiment builds: tools/build_variant.sh NAME "-DFLAG ..." file.hip [file.hip ...]
# recompiles the named sources with the extra flags and links them with the product's other objects into
# pytorch-models_amd/csrc/build/var_NAME/libpm_mi355x.so; run with PM_MI355X_LIB=<that path>.
set -e
cd "$(dirname "$0")/../pytorch-models_amd/csrc"
name=$1; flags=$2; shift 2
make -j8 >/dev/null
d=build/var_$name; mkdir -p $d
objs=""
for o in build/*.o; do
  b=$(basename $o .o); skip=0
  for f in "$@"; do [ "$b" == "$(basename $f .hip)" ] && skip=1; done
  [ $skip == 0 ] && objs="$objs $o"
done
for f in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 $flags -c $f -o $d/$(basename $f .hip).o
  objs="$objs $d/$(basename $f .hip).o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libpm_mi355x.so $objs
echo $PWD/$d/libpm_mi355x.so
