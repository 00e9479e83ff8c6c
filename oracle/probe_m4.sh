#!/bin/bash
# oracle/probe_m4.sh -- evidence, build container only (round 6; companion of probe_generic.sh): does the reference's covarion
# ("M4", Markov-modulated) code -- the only place that sets a state count other than 4 / 20 (mod->ns = n_o * n_h,
# src/init.c:6407) -- reach Update_Partial_Lk_Generic (src/lk.c:1332-1587) in ANY buildable, runnable form?
#
# It does not.  Four doors, each tried below with the reference's own objects (oracle/_ref) and sources where they lie:
#   1. the `-DM4` build itself (src/main.c:151-153, :440 `#elif (M4)` -> M4_main): src/cl.c does not COMPILE with -DM4 --
#      `io->mod->s_opt->opt_cov_alpha` / `opt_cov_delta` are no longer members of struct __Optimiz and `m4mod->alpha` /
#      `m4mod->delta` became scalar_dbl * (src/cl.c:709-739 assign doubles to them).  configure.ac has no m4 target either
#      (only the `m4.c m4.h` entries of the other programs' source lists survive in src/Makefile.am).
#   2. main.c's -DM4 call order (Init_Model; `if (use_m4mod) M4_Init_Model`, src/main.c:148-153 = src/m4.c:107-109) over the
#      cl.c that does compile (`--cov` sets use_m4mod, src/cl.c:753-757): mod->m4mod is NULL (src/init.c:705; nothing on the
#      command-line route allocates it) -> SIGSEGV inside M4_Init_Model, before Make_Tree_For_Lk.
#   3. the interactive route's own allocation (src/interface.c:110-118: ns *= n_h, M4_Make_Light, M4_Init_Model,
#      M4_Make_Complete -- in THAT order, i.e. M4_Init_Model writes m4mod->o_fq before M4_Make_Complete allocates it).  The
#      probe is charitable: it allocates first (M4_Make_Light + M4_Make_Complete), multiplies ns before Make_Model_Complete so
#      that the eigen system and the rate matrix are sized for n_o * n_h, then follows main.c.  Init_Model -> Update_Eigen on
#      the 12 x 12 system a 4-state model fills only a corner of: "imaginary eigenvectors" and a crash in the error print.
#   4. the same with ns put back to n_o for Init_Model (so that the nucleotide model initialises as it would without --cov):
#      Eigen() aborts on its own size assertion (the eigen struct was made for 12 states) -- still inside Init_Model.
# No likelihood function appears in any backtrace.  State counts other than 4 / 20 therefore have NO reference behaviour to pin
# (DESIGN.md section 8); what IS reachable and pinned of the generic loop is its arithmetic on 4- and 20-state data
# (`phyml --cov`, tests/golden/nucleic_cov_generic.phyg, PHYHIP_FLAG_GENERIC_LOOP), see probe_generic.sh.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
T=$(mktemp -d)
FL="-std=gnu99 -O1 -g -mavx2 -mfma -DHAVE_CONFIG_H -I$REF -I$REF/src -w -fPIC"
cat > $T/segv.c <<'EOC'
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void h(int s) { void *b[48]; int n = backtrace(b, 48); (void)s; backtrace_symbols_fd(b, n, 2); _exit(139); }
__attribute__((constructor)) static void init(void) { signal(SIGSEGV, h); signal(SIGABRT, h); }
EOC
cd $T
cp $REF/examples/nucleic nuc.phy; chmod 644 nuc.phy
show() { grep -oE "probe[0-9]*\(([A-Za-z_0-9]+)\+" $1 | sed 's/probe[0-9]*(//; s/+//' | tr '\n' ' '; echo; }
lkfun() { grep -cE 'Update_Partial_Lk|Lk_Core|\(Lk\+' $1 || true; }
ARGS="-i nuc.phy -d nt -m HKY85 -c 4 -a 1.0 -o n -b 0 --cov"

echo "== door 1: src/cl.c with -DM4"
set +e
gcc $FL -DM4 -c $REF/src/cl.c -o cl_m4.o 2> cl_m4.err; st=$?
echo "gcc exit status $st; $(grep -c 'error:' cl_m4.err) errors, first: $(grep -m1 'error:' cl_m4.err | sed "s|$REF/||")"
gcc $FL -DM4 -c $REF/src/main.c -o main_m4.o 2> /dev/null && echo "(src/main.c alone does compile with -DM4)"

# the three driver variants: this repo's ref_driver.c (main.c's call order) with the -DM4 lines of main.c added
python3 - "$HERE/ref_driver.c" <<'EOP'
import sys
s = open(sys.argv[1]).read()
s = s.replace('#include "optimiz.h"', '#include "optimiz.h"\n#include "m4.h"')
init = '  Init_Model(cdata, mod, io);\n'
assert init in s
door2 = s.replace(init, init + '  if (io->mod->use_m4mod) M4_Init_Model(mod->m4mod, cdata, mod); /* src/main.c:151-153 */\n', 1)
open('drv2.c', 'w').write(door2)
alloc = ('  if (io->mod->use_m4mod) { int nh = 3; io->mod->m4mod = M4_Make_Light(); M4_Make_Complete(nh, io->mod->ns, io->mod->m4mod);\n'
         '    io->mod->m4mod->delta->v = 0.7; io->mod->m4mod->alpha->v = 0.5; io->mod->ns *= nh; } /* src/interface.c:110-118, allocation first */\n')
mk = '  Make_Model_Complete(io->mod);\n'
assert mk in s
door3 = door2.replace(mk, alloc + mk, 1)
open('drv3.c', 'w').write(door3)
door4 = door3.replace(init, '  if (io->mod->use_m4mod) mod->ns = mod->m4mod->n_o;\n' + init, 1)
open('drv4.c', 'w').write(door4)
EOP
for d in 2 3 4; do
  gcc $FL -rdynamic drv$d.c segv.c "$HERE/_ref/libphyml_ref.a" -lm -o probe$d
  echo "== door $d"
  timeout 120 ./probe$d bench 1 -- $ARGS < /dev/null > out$d.txt 2>&1; st=$?
  echo -n "backtrace: "; show out$d.txt
  echo "exit status $st; likelihood functions in the backtrace: $(lkfun out$d.txt); REF_BENCH lines: $(grep -c REF_BENCH out$d.txt)"
done
echo "== control: the same command line without M4_Init_Model (the generic loop on 4 states, pinned)"
timeout 120 "$HERE/_ref/phyml_ref_driver" bench 1 -- $ARGS < /dev/null 2>&1 | grep -o 'REF_BENCH.\{0,120\}' | head -1
rm -rf $T
