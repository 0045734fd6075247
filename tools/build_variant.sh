#!/bin/bash
# An A/B build of ONE device unit: tools/build_variant.sh <name> <unit> "<-D switches>"
#   -> ocean-perception_amd/lib/libvehicle_pm_gpu_<name>.so = the shipped objects with <unit>.o recompiled under the switches
# (load it through PM_LIB).  Needs `make` first.
set -e
name=$1; unit=$2; defs=$3
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/ocean-perception_amd
mkdir -p $pkg/build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -Wno-pass-failed \
  $defs -I$root/include -I$pkg/csrc -c -o $pkg/build/variants/${unit}_$name.o $pkg/csrc/$unit.hip
objs=""
for o in $pkg/build/*.o; do
  b=$(basename $o .o)
  if [ "$b" = "$unit" ]; then objs="$objs $pkg/build/variants/${unit}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/lib/libvehicle_pm_gpu_$name.so $objs -lz
echo built libvehicle_pm_gpu_$name.so
