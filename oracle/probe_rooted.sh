#!/bin/bash
# oracle/probe_rooted.sh -- evidence, build container only: does the reference's own AVX build evaluate a ROOTED tree with
# tree->ignore_root == NO (the special cases of Set_All_Partial_Lk, src/lk.c:2988-3146)?
#
# It does not.  Lk(NULL) on such a tree calls Update_Partial_Lk(tree, n_root->b[1], n_root) (src/lk.c:534-535; also
# Update_All_Partial_Lk, :409-410); for d == n_root, Set_All_Partial_Lk returns n_v1 = NULL (:3017-3021), and
# AVX_Update_Partial_Lk dereferences n_v1->tax unconditionally (src/avx.c:451,457,462) -> segmentation fault.  The `phyml`
# program never gets there (ignore_root stays YES, src/init.c:145; rooted evaluation belongs to the PhyTime / PhyREX
# programs, which SURVEY section 2 puts out of scope), so the rooted special cases have no reference behaviour on the
# AVX path to pin an implementation against; the engine's host layer and the glue keep rejecting n_root != NULL.
#
# The probe is oracle/ref_driver.c with two lines added before Make_Tree_For_Lk (Add_Root on edge 0, ignore_root = NO) and
# a SIGSEGV handler that prints the backtrace; nothing of it is committed besides this recipe.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
T=$(mktemp -d)
sed 's|  Set_Both_Sides(o->both_sides ? YES : NO, tree);|  if (getenv("PROBE_ROOTED")) { Add_Root(tree->a_edges[0], tree); tree->ignore_root = NO; }\n  Set_Both_Sides(o->both_sides ? YES : NO, tree);|' "$HERE/ref_driver.c" > $T/probe.c
cat > $T/segv.c <<'EOC'
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void h(int s) { void *b[32]; int n = backtrace(b, 32); (void)s; backtrace_symbols_fd(b, n, 2); _exit(139); }
__attribute__((constructor)) static void init(void) { signal(SIGSEGV, h); }
EOC
gcc -std=gnu99 -O0 -g -rdynamic -mavx2 -mfma -DHAVE_CONFIG_H -I$REF -I$REF/src -w $T/probe.c $T/segv.c $REF/src/avx.c $REF/src/lk.c \
    -o $T/probe "$HERE/_ref/libphyml_ref.a" -lm
cp $REF/examples/nucleic $T/ && chmod 644 $T/nucleic
cd $T
echo "== unrooted (control)"; ./probe bench 1 -- -i nucleic -d nt -m GTR -c 4 -a 1.0 -o n -b 0 2>&1 | grep REF_BENCH | cut -c1-60
echo "== rooted, ignore_root = NO"; set +e
PROBE_ROOTED=1 ./probe bench 1 -- -i nucleic -d nt -m GTR -c 4 -a 1.0 -o n -b 0 2>&1 | grep -E "probe\(" | head -5
echo "exit status ${PIPESTATUS[0]}"
rm -rf $T
