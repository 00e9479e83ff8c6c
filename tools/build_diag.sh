#!/bin/bash
# Diagnostic build of the engine (-DPHYHIP_DIAG): timing-only kernel variants (PHYHIP_ABLATE, PHYHIP_NOLOADS -- results
# INVALID) and the first-generation kernels as A/B references.  Goes to phyml_amd/lib_diag; select it with
# PHYHIP_LIBDIR=phyml_amd/lib_diag.  The product library (phyml_amd/lib, __graft_entry__.build()) has none of this.
set -e
cd "$(dirname "$0")/.."
mkdir -p phyml_amd/lib_diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DPHYHIP_DIAG \
  -o phyml_amd/lib_diag/libphyhip.so phyml_amd/csrc/phyhip.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
gcc -std=gnu99 -O2 -fPIC -shared -mfma -o phyml_amd/lib_diag/libphyhip_lk.so phyml_amd/csrc/host/phl_lk.c \
  -Lphyml_amd/lib_diag -lphyhip -lm -Wl,-rpath,'$ORIGIN'
