#!/bin/bash
# Diagnostic build of the engine (-DPHYHIP_DIAG): timing-only kernel variants (PHYHIP_ABLATE, PHYHIP_NOLOADS -- results
# INVALID) and the first-generation kernels as A/B references.  Goes to phyml_amd/lib_diag; select it with
# PHYHIP_LIBDIR=phyml_amd/lib_diag.  The product library (phyml_amd/lib, __graft_entry__.build()) has none of this.
exec "$(dirname "$0")/build_variant.sh" diag -DPHYHIP_DIAG
