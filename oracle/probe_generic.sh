#!/bin/bash
# oracle/probe_generic.sh -- evidence, build container only: can the reference's `phyml` program reach
# Update_Partial_Lk_Generic (src/lk.c:1332-1587, dispatched from Update_Partial_Lk at src/lk.c:1317,1322 when the state count
# is neither 4 nor 20) and Init_Tips_At_One_Site_Generic_Float (src/lk.c:210-237)?
#
# It cannot.  There are two doors to io->datatype = GENERIC in this build -- `-d generic` (src/cl.c:929-932) and a NEXUS file
# with `format datatype=standard` (src/nexus.c:241-251); the covarion models that multiply the state count need the separate
# `m4` program (-DM4).  The probe runs the reference's own objects (oracle/_ref, built by oracle/Makefile from
# /root/reference/src) on a three-state alignment through both doors and prints the backtrace of what happens:
#   * `-d generic`: segmentation fault in Eigen() <- Update_Eigen() <- Init_Model() -- the model set-up of a generic
#     alphabet, before Make_Tree_For_Lk, before any likelihood function;
#   * NEXUS datatype=standard: Init_Model() -> Update_Eigen() -> Eigen() -> elemhess() never returns (killed after 20 s).
# (A named -m model cannot be added: every one of them resets the data type to NT or AA, src/cl.c:945-1100; `-o n` without
# `-m` dereferences a NULL r_mat in Read_Command_Line, src/cl.c:1882, whatever the data type -- hence the bare command.)
# So no generic STATE COUNT has a reference behaviour to pin an implementation against; the engine
# answers PHYHIP_ERROR_NO_IMPLEMENTATION for state counts other than 4 and 20.
# The third door (round 4) does open: `--cov` (src/cl.c:753-757) sets mod->use_m4mod, M4_Init_Model is compiled out of the
# `phyml` program (src/main.c:151, -DM4 only), so Update_Partial_Lk (src/lk.c:1303-1324) sends the unchanged 4-state data
# through Update_Partial_Lk_Generic -- the probe prints its lnL next to the default path's (equal to the last digit on
# examples/nucleic, at half the speed).  That behaviour IS pinned: tests/golden/nucleic_cov_generic.phyg
# (tests/golden/make_cov.py), oracle arith = 2, PHYHIP_FLAG_GENERIC_LOOP.
# Control: the nucleotide recoding of the same alignment runs to its likelihood with the same driver.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
T=$(mktemp -d)
cat > $T/segv.c <<'EOC'
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void h(int s) { void *b[48]; int n = backtrace(b, 48); (void)s; backtrace_symbols_fd(b, n, 2); _exit(139); }
__attribute__((constructor)) static void init(void) { signal(SIGSEGV, h); signal(SIGABRT, h); }
EOC
gcc -std=gnu99 -O0 -g -rdynamic -mavx2 -mfma -DHAVE_CONFIG_H -I$REF -I$REF/src -w "$HERE/ref_driver.c" $T/segv.c \
    -o $T/probe "$HERE/_ref/libphyml_ref.a" -lm
cat > $T/gen.phy <<'EOA'
5 12
t1    012012012012
t2    012112012002
t3    002012112012
t4    012012012212
t5    112012002012
EOA
tr '012' 'ACG' < $T/gen.phy | sed '1s/.*/5 12/' > $T/nt.phy
{ printf '#NEXUS\nbegin data;\ndimensions ntax=5 nchar=12;\nformat datatype=standard symbols="012";\nmatrix\n'; tail -n +2 $T/gen.phy; printf ';\nend;\n'; } > $T/gen.nex
cd $T
show() { grep -oE "probe\(([A-Za-z_0-9]+)\+" $1 | sed 's/probe(//; s/+//' | tr '\n' ' '; echo; }
echo "== control: -d nt on the recoded alignment"
timeout 60 ./probe bench 1 -- -i nt.phy -d nt -b 0 < /dev/null 2>&1 | grep -o 'REF_BENCH.\{0,80\}' | head -1
set +e
echo "== -d generic"
timeout 60 ./probe bench 1 -- -i gen.phy -d generic -b 0 < /dev/null > out1.txt 2>&1; st=$?
echo -n "backtrace: "; show out1.txt
echo "exit status $st; likelihood functions in the backtrace: $(grep -cE 'Update_Partial_Lk|Lk_Core|\(Lk\+' out1.txt)"
echo "== NEXUS, datatype=standard symbols=\"012\" (aborted after 20 s)"
timeout -s ABRT 20 ./probe bench 1 -- -i gen.nex -b 0 < /dev/null > out2.txt 2>&1; st=$?
echo -n "backtrace: "; show out2.txt
echo "exit status $st; likelihood functions in the backtrace: $(grep -cE 'Update_Partial_Lk|Lk_Core|\(Lk\+' out2.txt)"
echo "== --cov on examples/nucleic (GTR+G4): the generic loop on 4-state data, against the default path"
cp $REF/examples/nucleic nuc.phy; chmod 644 nuc.phy
timeout 120 ./probe bench 1 -- -i nuc.phy -d nt -m GTR -c 4 -a 1.0 -o n -b 0 < /dev/null 2>&1 | grep -o 'REF_BENCH.\{0,160\}' | head -1
timeout 120 ./probe bench 1 -- -i nuc.phy -d nt -m GTR -c 4 -a 1.0 -o n -b 0 --cov < /dev/null 2>&1 | grep -o 'REF_BENCH.\{0,160\}' | head -1
rm -rf $T
