#!/bin/bash
# where the fused Update_Eigen_Lr stops paying: phyml_amd/lib_fz (tools/build_variant.sh fz -DPHYHIP_FUSE_EIGEN_MAX=100000000: always
# fused) against phyml_amd/lib_nf (-DPHYHIP_FUSE_EIGEN_MAX=0: never) on one box, resident evaluators off and on
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for res in 0 1; do for v in lib_fz lib_nf; do echo "== $v PHYHIP_RESIDENT=$res"; PHYHIP_RESIDENT=$res PHYHIP_LIBDIR=$R/phyml_amd/$v timeout 300 python tools/fuse_eigen_cross.py 2>/dev/null; done; done
