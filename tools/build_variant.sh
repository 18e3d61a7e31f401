#!/bin/bash
# Usage: tools/build_variant.sh NAME "<extra -D flags>" [TU ...]
# Builds ldsr_amd/libldsr_hip_NAME.so for same-box A/B runs (tools/ab.sh): the listed translation
# units (default: em_scan_L16 kernels_scan) are recompiled with the extra flags, every other
# object is taken from the normal build (run `make -C ldsr_amd/csrc` first).
set -e
name=$1; flags=$2; shift 2
tus=${@:-em_scan_L16 kernels_scan}
cd "$(dirname "$0")/../ldsr_amd/csrc"
mkdir -p /tmp/ldsr_var_$name
objs=""
for o in *.o; do
  b=${o%.o}
  if echo " $tus " | grep -q " $b "; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 --offload-compress --offload-compression-level=19 -Wall -Wno-unused-function $flags -c $b.hip -o /tmp/ldsr_var_$name/$b.o &
    objs="$objs /tmp/ldsr_var_$name/$b.o"
  else
    objs="$objs $o"
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 --offload-compress -o ../libldsr_hip_$name.so $objs
echo built ldsr_amd/libldsr_hip_$name.so
