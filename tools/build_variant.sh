#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...] : build the engine with extra -D flags into phyml_amd/lib_<name>
# (select with PHYHIP_LIBDIR) -- A/B timing of compile-time choices in ONE gpurun call.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p phyml_amd/lib_$name/obj
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC"
/opt/rocm/bin/hipcc $F "$@" -c -o phyml_amd/lib_$name/obj/phyhip.o phyml_amd/csrc/phyhip.hip &
/opt/rocm/bin/hipcc $F -mllvm -disable-machine-licm "$@" -c -o phyml_amd/lib_$name/obj/phyhip_big.o phyml_amd/csrc/phyhip_big.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o phyml_amd/lib_$name/libphyhip.so phyml_amd/lib_$name/obj/phyhip.o \
  phyml_amd/lib_$name/obj/phyhip_big.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
gcc -std=gnu99 -O2 -fPIC -shared -mfma -o phyml_amd/lib_$name/libphyhip_lk.so phyml_amd/csrc/host/phl_lk.c \
  -Lphyml_amd/lib_$name -lphyhip -lm -Wl,-rpath,'$ORIGIN'
