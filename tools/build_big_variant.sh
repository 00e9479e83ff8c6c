#!/bin/bash
# tools/build_big_variant.sh <name> [extra hipcc flags...] : as build_variant.sh, for compile-time choices that only touch
# phyhip_big.hip (the resident kernels): that unit alone is recompiled, the product's other objects (phyml_amd/lib/obj) are linked in.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p phyml_amd/lib_$name/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -disable-machine-licm "$@" -c -o phyml_amd/lib_$name/obj/phyhip_big.o phyml_amd/csrc/phyhip_big.hip
objs=""
for u in phyhip phyhip_queue phyhip_resident phyhip_eigen phyhip_mixture phyhip_shard; do objs="$objs phyml_amd/lib/obj/$u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o phyml_amd/lib_$name/libphyhip.so $objs phyml_amd/lib_$name/obj/phyhip_big.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
gcc -std=gnu99 -O2 -fPIC -shared -mfma -o phyml_amd/lib_$name/libphyhip_lk.so phyml_amd/csrc/host/phl_lk.c -Lphyml_amd/lib_$name -lphyhip -lm -Wl,-rpath,'$ORIGIN'
