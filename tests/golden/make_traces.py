#!/usr/bin/env python3
"""Record surface-call traces from REAL PhyML tree searches (tests/golden/trace_*.phyg).

Runs only in the build container (needs /root/reference and `make -C oracle ref`):
    python tests/golden/make_traces.py
oracle/_ref/phyml_trace_driver runs the unmodified reference (SPR search + branch-length optimisation on the
reference's own example alignments) with the likelihood surface interposed, and writes the first N records of
the call stream -- buffer-level operations plus the scalar every call returned -- together with the inputs a
replay needs (model block, tip state sets, pattern weights).  The outputs are data only.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "phyml_trace_driver")
REF = os.environ.get("PHYML_REF", "/root/reference")

TRACES = {
    # name: (example file, max records, driver opts, phyml args)
    "trace_nucleic_spr": ("nucleic", 12000, ["--gtr-rr", "1,2.5,0.8,1.2,3.0,1"],
                          ["-d", "nt", "-m", "GTR", "-f", "0.3,0.2,0.2,0.3", "-c", "4", "-a", "0.8", "-s", "SPR", "-o", "tl", "-b", "0",
                           "--r_seed", "1"]),
    "trace_proteic_spr": ("proteic", 9000, [],
                          ["-d", "aa", "-m", "LG", "-f", "m", "-c", "4", "-a", "1.0", "-s", "SPR", "-o", "tl", "-b", "0", "--r_seed", "1"]),
}


# A LARGE real search (round 4): 200 taxa, synthetic alignment (phyml_amd/synth.py: seeded integer-hash generator, 260 sites so
# that the reference's `int` slab holds it), PhyML's own SPR search from its BioNJ tree.  The first 46 000 surface calls: the
# initial Lk, four rounds of Br_Len_Opt over the 397 edges (3 570 Update_Eigen_Lr, 12 757 dLk), then ~9 000 records of the SPR
# phase proper (src/spr.c:149,813: path updates, regraft candidates).  tests/test_gpu_cfg5.py replays this stream at 100 000
# patterns: cfg5's pin with a recorded instead of a seeded call stream.
SYNTH_TRACES = {
    # name: (n_otu, n_sites, seed, lmin, lmax, max records, driver opts, phyml args)
    "trace_synth200_spr": (200, 260, 41, 0.02, 0.12, 46000, ["--gtr-rr", "1,2.5,0.8,1.2,3.0,1"],
                           ["-d", "nt", "-m", "GTR", "-f", "0.3,0.2,0.2,0.3", "-c", "4", "-a", "0.8", "-s", "SPR", "-o", "tl", "-b", "0",
                            "--r_seed", "1"]),
}

GLUE = os.path.join(ROOT, "oracle", "_ref", "phyml_glue_driver")
SEARCHES = {
    # name: (example file, driver opts, phyml args) -- SPR topology search + branch lengths, model fixed
    "search_nucleic_spr": ("nucleic", ["--gtr-rr", "1,2.5,0.8,1.2,3.0,1"],
                           ["-d", "nt", "-m", "GTR", "-f", "0.3,0.2,0.2,0.3", "-c", "4", "-a", "0.8", "-s", "SPR", "-o", "tl", "-b", "0",
                            "--r_seed", "1"]),
    "search_proteic_spr": ("proteic", [],
                           ["-d", "aa", "-m", "LG", "-f", "m", "-c", "4", "-a", "1.0", "-s", "SPR", "-o", "tl", "-b", "0", "--r_seed", "1"]),
}


def main():
    if not os.path.exists(DRIVER):
        raise SystemExit("build oracle/_ref first: make -C oracle ref")
    tmp = tempfile.mkdtemp(prefix="traces_")
    only = set(sys.argv[1:])  # (names on the command line: regenerate only those)
    for name, (example, nrec, dopts, pargs) in TRACES.items():
        if only and name not in only:
            continue
        shutil.copy(os.path.join(REF, "examples", example), tmp)
        os.chmod(os.path.join(tmp, example), 0o644)
        out = os.path.join(HERE, name + ".phyg")
        r = subprocess.run([DRIVER, out, str(nrec)] + dopts + ["--", "-i", example] + pargs, cwd=tmp,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        m = re.search(r"TRACE_DRIVER records=.*", r.stdout)
        if r.returncode != 0 or not m:
            print(r.stdout[-3000:])
            raise SystemExit(f"trace driver failed for {name}")
        print(f"{name:22s} {m.group(0)}  {os.path.getsize(out) / 1024:.0f} KiB")

    sys.path.insert(0, ROOT)
    from phyml_amd import synth
    for name, (n, sites, seed, lmin, lmax, nrec, dopts, pargs) in SYNTH_TRACES.items():
        if only and name not in only:
            continue
        tree = synth.random_tree(n, seed, lmin, lmax)
        st = synth.simulate_states(tree, sites, 4, seed)
        synth.write_phylip(os.path.join(tmp, name + ".phy"), tree.names, synth.states_to_chars(st, 4))
        out = os.path.join(HERE, name + ".phyg")
        r = subprocess.run([DRIVER, out, str(nrec)] + dopts + ["--", "-i", name + ".phy"] + pargs, cwd=tmp,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        m = re.search(r"TRACE_DRIVER records=.*", r.stdout)
        if r.returncode != 0 or not m:
            print(r.stdout[-3000:])
            raise SystemExit(f"trace driver failed for {name}")
        print(f"{name:22s} {m.group(0)}  {os.path.getsize(out) / 1024:.0f} KiB")
    if only:
        shutil.rmtree(tmp, ignore_errors=True)
        return 0

    # --- whole searches: the CPU-only run of the command lines tests/test_gpu_search.py repeats on the GPU ----------
    import json
    expected = {}
    for name, (example, dopts, pargs) in SEARCHES.items():
        # the example alignments travel as data files (the GPU box has no /root/reference)
        shutil.copy(os.path.join(REF, "examples", example), os.path.join(HERE, "examples_" + example + ".phy"))
        os.chmod(os.path.join(HERE, "examples_" + example + ".phy"), 0o644)
        env = dict(os.environ, GLUE_MODE="host")
        r = subprocess.run([GLUE] + dopts + ["--", "-i", example] + pargs, cwd=tmp, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
        if r.returncode != 0 or not m:
            print(r.stdout[-3000:])
            raise SystemExit(f"glue driver (host mode) failed for {name}")
        info = json.loads(m.group(1))
        expected[name] = dict(example=example, driver_opts=dopts, phyml_args=pargs, lnL_init=info["lnL_init"],
                              lnL_final=info["lnL_final"], tree=info["tree"], calls=info["calls"], cpu_seconds=info["seconds"])
        print(f"{name:22s} lnL {info['lnL_init']:.6f} -> {info['lnL_final']:.6f}  {info['calls']}  {info['seconds']:.1f} s (1 core, this container)")
    json.dump(expected, open(os.path.join(HERE, "search_expected.json"), "w"), indent=1, sort_keys=True)

    # --- mixture golden vector: the first MIXT_Lk of the LG4X example with every optimisation switched off ----------
    mixt = os.path.join(ROOT, "oracle", "_ref", "phyml_mixt_driver")
    os.makedirs(os.path.join(tmp, "examples", "lg4x"), exist_ok=True); os.makedirs(os.path.join(tmp, "run"), exist_ok=True)
    for f in ("X1.mat", "X2.mat", "X3.mat", "X4.mat"):
        shutil.copy(os.path.join(REF, "examples", "lg4x", f), os.path.join(HERE, "lg4x", f))
        os.chmod(os.path.join(HERE, "lg4x", f), 0o644)
        shutil.copy(os.path.join(HERE, "lg4x", f), os.path.join(tmp, "examples", "lg4x", f))
    shutil.copy(os.path.join(REF, "examples", "proteic"), os.path.join(tmp, "examples", "proteic"))
    xml = open(os.path.join(HERE, "lg4x", "lg4x_check.xml")).read()
    xml = xml.replace('optimise.freerates="yes"', 'optimise.freerates="no"').replace('optimise.lens="yes"', 'optimise.lens="no"')
    open(os.path.join(tmp, "examples", "lg4x", "fixed.xml"), "w").write(xml)
    out = os.path.join(HERE, "mixture_lg4x.phyg")
    r = subprocess.run([mixt, out, "0", "--", "--xml=../examples/lg4x/fixed.xml"], cwd=os.path.join(tmp, "run"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    m = re.search(r"MIXT_DRIVER .*", r.stdout)
    if r.returncode != 0 or not m:
        print(r.stdout[-3000:])
        raise SystemExit("mixture driver failed")
    print(f"{'mixture_lg4x':22s} {m.group(0)}  {os.path.getsize(out) / 1024:.0f} KiB")
    # ... and one MIXT_dLk call of the real analysis (41st of the first branch-length round)
    shutil.copy(os.path.join(HERE, "lg4x", "lg4x_check.xml"), os.path.join(tmp, "examples", "lg4x", "lg4x_check.xml"))
    out = os.path.join(HERE, "mixture_lg4x_dlk.phyg")
    r = subprocess.run([mixt, out, "40", "dlk", "--", "--xml=../examples/lg4x/lg4x_check.xml"], cwd=os.path.join(tmp, "run"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    m = re.search(r"MIXT_DRIVER .*", r.stdout)
    if r.returncode != 0 or not m:
        print(r.stdout[-3000:])
        raise SystemExit("mixture driver (dlk) failed")
    print(f"{'mixture_lg4x_dlk':22s} {m.group(0)}  {os.path.getsize(out) / 1024:.0f} KiB")
    # --- the same two dumps for a four-class NUCLEOTIDE mixture (tests/golden/ntmix/nt4_check.xml: this repo's own input) ---
    os.makedirs(os.path.join(tmp, "examples", "ntmix"), exist_ok=True)
    shutil.copy(os.path.join(REF, "examples", "nucleic"), os.path.join(tmp, "examples", "nucleic"))
    os.chmod(os.path.join(tmp, "examples", "nucleic"), 0o644)
    xml = open(os.path.join(HERE, "ntmix", "nt4_check.xml")).read()
    open(os.path.join(tmp, "examples", "ntmix", "nt4_check.xml"), "w").write(xml)
    open(os.path.join(tmp, "examples", "ntmix", "fixed.xml"), "w").write(
        xml.replace('optimise.freerates="yes"', 'optimise.freerates="no"').replace('optimise.lens="yes"', 'optimise.lens="no"'))
    for name, argv in (("mixture_nt4", ["0", "--", "--xml=../examples/ntmix/fixed.xml"]),
                       ("mixture_nt4_dlk", ["40", "dlk", "--", "--xml=../examples/ntmix/nt4_check.xml"])):
        out = os.path.join(HERE, name + ".phyg")
        r = subprocess.run([mixt, out] + argv, cwd=os.path.join(tmp, "run"), stdin=subprocess.DEVNULL, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        m = re.search(r"MIXT_DRIVER .*", r.stdout)
        if r.returncode != 0 or not m:
            print(r.stdout[-3000:])
            raise SystemExit(f"mixture driver failed for {name}")
        print(f"{name:22s} {m.group(0)}  {os.path.getsize(out) / 1024:.0f} KiB")
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
