#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...] : build the engine with extra -D flags into phyml_amd/lib_<name>
# (select with PHYHIP_LIBDIR) -- A/B timing of compile-time choices in ONE gpurun call.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p phyml_amd/lib_$name/obj
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC"
objs=""
for u in phyhip phyhip_queue phyhip_resident phyhip_eigen phyhip_mixture phyhip_shard; do
  /opt/rocm/bin/hipcc $F "$@" -c -o phyml_amd/lib_$name/obj/$u.o phyml_amd/csrc/$u.hip &
  objs="$objs phyml_amd/lib_$name/obj/$u.o"
done
/opt/rocm/bin/hipcc $F -mllvm -disable-machine-licm "$@" -c -o phyml_amd/lib_$name/obj/phyhip_big.o phyml_amd/csrc/phyhip_big.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o phyml_amd/lib_$name/libphyhip.so $objs \
  phyml_amd/lib_$name/obj/phyhip_big.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
gcc -std=gnu99 -O2 -fPIC -shared -mfma -o phyml_amd/lib_$name/libphyhip_lk.so phyml_amd/csrc/host/phl_lk.c \
  -Lphyml_amd/lib_$name -lphyhip -lm -Wl,-rpath,'$ORIGIN'
