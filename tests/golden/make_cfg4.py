#!/usr/bin/env python3
"""Expected log-likelihoods of BASELINE configs[3] (cfg4: 100 taxa x 1 000 000 nt patterns) from the REAL reference.

The unmodified reference cannot allocate 1 M patterns in one piece (its slab size is an `int`, src/make.c:96-104),
but lnL is additive over pattern shards once the model is data-independent (fixed frequencies), so the reference is
run on the eight contiguous 125 000-pattern shards the 8-GPU configuration uses and the per-shard values plus their
sum go into tests/golden/manifest.json ("cfg4_nt_100x1M").  Runs only in the build container:
    make -C oracle ref && python tests/golden/make_cfg4.py
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
import numpy as np  # noqa: E402
from phyml_amd import synth, workloads  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "phyml_ref_driver")


def main():
    man_path = os.path.join(HERE, "manifest.json")
    man = json.load(open(man_path))
    cfg = workloads.CONFIGS["cfg4_nt_100x1M"]
    n, P, seed, G = cfg["n_otu"], cfg["n_pattern"], cfg["seed"], 8
    tree = synth.random_tree(n, seed, 0.02, 0.15)
    tmp = tempfile.mkdtemp(prefix="cfg4_")
    tre = os.path.join(tmp, "t.nwk")
    open(tre, "w").write(tree.to_newick() + "\n")
    shards, cks = [], []
    for g in range(G):
        lo, hi = g * P // G, (g + 1) * P // G
        st = synth.simulate_states(tree, hi - lo, 4, seed, site_offset=lo)
        ali = os.path.join(tmp, f"s{g}.phy")
        synth.write_phylip(ali, tree.names, synth.states_to_chars(st, 4))
        out = subprocess.run([DRIVER, "bench", "1", "--gtr-rr", man["gtr_rr"], "--", "-i", ali, "-u", tre, "-d", "nt", "-m", "GTR",
                              "-f", man["nt_freq"], "-c", "4", "-a", "1.0", "-o", "n", "-b", "0", "--no_colalias"],
                             cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        info = json.loads(re.search(r"REF_BENCH (\{.*\})", out).group(1))
        assert info["n_pattern"] == hi - lo, info
        shards.append(info["lnL"]); cks.append(synth.states_checksum(st))
        print(f"shard {g}: [{lo},{hi}) lnL={info['lnL']!r} ref {info['site_updates_per_s'] / 1e6:.2f} M site-updates/s", flush=True)
        os.remove(ali)
    total = 0.0
    for v in shards:
        total += v
    man["expected"]["cfg4_nt_100x1M"] = dict(n_otu=n, n_pattern=P, ns=4, seed=seed, lmin=0.02, lmax=0.15, shards=G,
                                             shard_lnL=shards, shard_checksum=cks, lnL=total)
    json.dump(man, open(man_path, "w"), indent=1, sort_keys=True)
    print("cfg4 total lnL", repr(total))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
