#!/usr/bin/env python3
"""Regenerate the golden vectors under tests/golden/ from the REAL reference.

Runs only where /root/reference and oracle/_ref/phyml_ref_driver exist (the build container):
    make -C oracle ref && python tests/golden/make_golden.py
The outputs are data only (inputs + expected outputs of the reference's Lk()/dLk()/partials);
no reference source travels.  Synthetic inputs come from phyml_amd/synth.py (integer-hash
generator), so the GPU box regenerates identical alignments from the seeds recorded here.
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import phyg  # noqa: E402
from phyml_amd import synth  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "phyml_ref_driver")
REF = os.environ.get("PHYML_REF", "/root/reference")

GTR_RR = "1,2.5,0.8,1.2,3.0,1"
NT_FREQ = "0.3,0.2,0.2,0.3"


def run(args, cwd):
    r = subprocess.run([DRIVER] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout[-3000:])
        raise SystemExit(f"driver failed: {args}")
    return r.stdout


def dump(name, drv_opts, phyml_args, cwd):
    out = os.path.join(HERE, name + ".phyg")
    txt = run(["dump", out] + drv_opts + ["--"] + phyml_args, cwd)
    m = re.search(r"REF_DRIVER lnL=(\S+)", txt)
    print(f"{name:28s} lnL={m.group(1)}  {os.path.getsize(out) / 1024:.0f} KiB")
    return float(m.group(1))


def synth_inputs(cwd, tag, n_otu, n_sites, ns, seed, lmin, lmax):
    tree = synth.random_tree(n_otu, seed, lmin, lmax)
    st = synth.simulate_states(tree, n_sites, ns, seed)
    ali = os.path.join(cwd, tag + ".phy")
    tre = os.path.join(cwd, tag + ".nwk")
    synth.write_phylip(ali, tree.names, synth.states_to_chars(st, ns))
    open(tre, "w").write(tree.to_newick() + "\n")
    return ali, tre, synth.states_checksum(st)


def main():
    if not os.path.exists(DRIVER):
        raise SystemExit("build oracle/_ref first: make -C oracle ref")
    tmp = tempfile.mkdtemp(prefix="golden_")
    for f in ("nucleic", "proteic"):
        shutil.copy(os.path.join(REF, "examples", f), tmp)
        os.chmod(os.path.join(tmp, f), 0o644)
    manifest = {}

    # --- the reference's own example files (known answers: BASELINE.md §2) ---------------------
    manifest["nucleic_gtr_g4"] = dump("nucleic_gtr_g4", ["--full-edges", "3", "--eigen-edges", "2"],
                                      ["-i", "nucleic", "-d", "nt", "-m", "GTR", "-c", "4", "-a", "1.0", "-o", "n", "-b", "0"], tmp)
    manifest["proteic_lg_g4"] = dump("proteic_lg_g4", ["--full-edges", "2", "--eigen-edges", "1", "--pmat-edges", "8"],
                                     ["-i", "proteic", "-d", "aa", "-m", "LG", "-c", "4", "-a", "1.0", "-o", "n", "-b", "0"], tmp)
    # non-trivial GTR, fixed frequencies, +I, alpha 0.7
    manifest["nucleic_gtr_g4_inv"] = dump("nucleic_gtr_g4_inv", ["--gtr-rr", GTR_RR, "--full-edges", "1", "--eigen-edges", "1", "--pmat-edges", "8"],
                                          ["-i", "nucleic", "-d", "nt", "-m", "GTR", "-f", NT_FREQ, "-c", "4", "-a", "0.7", "-v", "0.2", "-o", "n", "-b", "0"], tmp)
    # bootstrap-like zero-weight patterns
    manifest["nucleic_zero_w"] = dump("nucleic_zero_w", ["--zero-weights", "5", "--full-edges", "1", "--eigen-edges", "1", "--pmat-edges", "8"],
                                      ["-i", "nucleic", "-d", "nt", "-m", "GTR", "-c", "4", "-a", "1.0", "-o", "n", "-b", "0"], tmp)

    # --- synthetic, deep trees that force the 2^256 rescaling rule -------------------------------
    synth_meta = {}
    ali, tre, ck = synth_inputs(tmp, "nt300", 300, 40, 4, 7, 0.05, 0.4)
    synth_meta["synth_nt_300x40"] = dict(n_otu=300, n_sites=40, ns=4, seed=7, lmin=0.05, lmax=0.4, checksum=ck)
    manifest["synth_nt_300x40"] = dump("synth_nt_300x40", ["--gtr-rr", GTR_RR, "--full-edges", "2", "--eigen-edges", "1", "--pmat-edges", "8"],
                                       ["-i", ali, "-u", tre, "-d", "nt", "-m", "GTR", "-f", NT_FREQ, "-c", "4", "-a", "1.0", "-o", "n", "-b", "0", "--no_colalias"], tmp)
    ali, tre, ck = synth_inputs(tmp, "aa90", 90, 24, 20, 8, 0.05, 0.4)
    synth_meta["synth_aa_90x24"] = dict(n_otu=90, n_sites=24, ns=20, seed=8, lmin=0.05, lmax=0.4, checksum=ck)
    manifest["synth_aa_90x24"] = dump("synth_aa_90x24", ["--full-edges", "2", "--eigen-edges", "1", "--pmat-edges", "4"],
                                      ["-i", ali, "-u", tre, "-d", "aa", "-m", "LG", "-f", "m", "-c", "4", "-a", "1.0", "-o", "n", "-b", "0", "--no_colalias"], tmp)

    # --- data-independent model blocks for the BASELINE configs ---------------------------------
    ali, tre, _ = synth_inputs(tmp, "m_nt", 6, 50, 4, 3, 0.05, 0.2)
    dump("model_gtr_g4", ["--gtr-rr", GTR_RR, "--model-only"],
         ["-i", ali, "-u", tre, "-d", "nt", "-m", "GTR", "-f", NT_FREQ, "-c", "4", "-a", "1.0", "-o", "n", "-b", "0", "--no_colalias"], tmp)
    ali, tre, _ = synth_inputs(tmp, "m_aa", 6, 50, 20, 3, 0.05, 0.2)
    dump("model_lg_g4", ["--model-only"],
         ["-i", ali, "-u", tre, "-d", "aa", "-m", "LG", "-f", "m", "-c", "4", "-a", "1.0", "-o", "n", "-b", "0", "--no_colalias"], tmp)

    # --- expected lnL of the BASELINE configs (reference AVX path on the regenerable inputs) ----
    expected = {}
    cfgs = {
        # name: (n_otu, patterns, ns, seed, phyml model args, driver opts)
        "cfg2_nt_100x50k": (100, 50000, 4, 1, ["-d", "nt", "-m", "GTR", "-f", NT_FREQ], ["--gtr-rr", GTR_RR]),
        "cfg3_aa_200x10k": (200, 10000, 20, 2, ["-d", "aa", "-m", "LG", "-f", "m"], []),
        "small_nt_24x2000": (24, 2000, 4, 5, ["-d", "nt", "-m", "GTR", "-f", NT_FREQ], ["--gtr-rr", GTR_RR]),
        "small_aa_16x600": (16, 600, 20, 6, ["-d", "aa", "-m", "LG", "-f", "m"], []),
    }
    for name, (n, P, ns, seed, margs, dopts) in cfgs.items():
        ali, tre, ck = synth_inputs(tmp, name, n, P, ns, seed, 0.02, 0.15)
        txt = run(["bench", "1"] + dopts + ["--"] + ["-i", ali, "-u", tre] + margs +
                  ["-c", "4", "-a", "1.0", "-o", "n", "-b", "0", "--no_colalias"], tmp)
        m = re.search(r"REF_BENCH (\{.*\})", txt)
        info = json.loads(m.group(1))
        expected[name] = dict(n_otu=n, n_pattern=P, ns=ns, seed=seed, lmin=0.02, lmax=0.15, checksum=ck,
                              lnL=info["lnL"], ref_site_updates_per_s=info["site_updates_per_s"])
        print(f"{name:28s} lnL={info['lnL']!r}  ref {info['site_updates_per_s'] / 1e6:.2f} M site-updates/s (this container, 1 core)")

    json.dump(dict(lnL=manifest, synthetic=synth_meta, expected=expected, gtr_rr=GTR_RR, nt_freq=NT_FREQ),
              open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
