#!/usr/bin/env python3
"""tests/golden/nucleic_cov_generic.phyg: the reference's GENERIC partial-likelihood loop (Update_Partial_Lk_Generic,
src/lk.c:1332-1587) on the one door through which the `phyml` program reaches it with a likelihood at the end:
`--cov` (src/cl.c:753-757) sets mod->use_m4mod, and Update_Partial_Lk (src/lk.c:1303-1324) then sends the 4-state data through
the generic loop instead of the AVX / SSE kernels (M4_Init_Model is compiled only into the separate `m4` program, src/main.c:151:
the state count stays 4).  Same command as nucleic_gtr_g4_inv + --cov; dumped with the same driver.
Build container only:  make -C oracle ref && python tests/golden/make_cov.py"""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    tmp = tempfile.mkdtemp(prefix="golden_cov_")
    shutil.copy(os.path.join(mg.REF, "examples", "nucleic"), tmp)
    os.chmod(os.path.join(tmp, "nucleic"), 0o644)
    lnl = mg.dump("nucleic_cov_generic", ["--gtr-rr", mg.GTR_RR, "--full-edges", "1", "--eigen-edges", "1", "--pmat-edges", "8"],
                  ["-i", "nucleic", "-d", "nt", "-m", "GTR", "-f", mg.NT_FREQ, "-c", "4", "-a", "0.7", "-v", "0.2", "-o", "n", "-b", "0", "--cov"], tmp)
    print("lnL", repr(lnl))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
