"""Synthetic workloads of BASELINE.json (`configs`) built from the seeded generator and the committed
data-independent model blocks (tests/golden/model_*.phyg, dumped from the reference: pi, Gamma rates and
weights, eigen system).  Used by bench.py and the full-size GPU tests."""
from __future__ import annotations

import json
import os

import numpy as np

from . import phyg, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# name -> (n_otu, n_pattern, ns, seed, model block)
CONFIGS = {
    "cfg2_nt_100x50k": dict(n_otu=100, n_pattern=50000, ns=4, seed=1, model="model_gtr_g4"),
    "cfg3_aa_200x10k": dict(n_otu=200, n_pattern=10000, ns=20, seed=2, model="model_lg_g4"),
    "cfg4_nt_100x125k": dict(n_otu=100, n_pattern=125000, ns=4, seed=1, model="model_gtr_g4"),
    "cfg4_nt_100x1M": dict(n_otu=100, n_pattern=1000000, ns=4, seed=1, model="model_gtr_g4"),
    "cfg5_nt_500x100k": dict(n_otu=500, n_pattern=100000, ns=4, seed=9, model="model_gtr_g4"),
    "small_nt_24x2000": dict(n_otu=24, n_pattern=2000, ns=4, seed=5, model="model_gtr_g4"),
    "small_aa_16x600": dict(n_otu=16, n_pattern=600, ns=20, seed=6, model="model_lg_g4"),
}


def manifest():
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))


def model_block(name):
    return phyg.load(os.path.join(GOLDEN, name + ".phyg"))


def make(name, n_pattern=None, seed=None, pattern_offset=0):
    """Returns dict(tree=EdgeTree, states=uint8[n_otu,P], model=block, cfg=...).  With n_pattern/seed
    overrides the same tree is kept (seed of the tree = config seed) and only the columns change; pattern_offset
    selects columns [pattern_offset, pattern_offset + n_pattern) of the configuration's alignment (a pattern shard)."""
    cfg = dict(CONFIGS[name])
    tree = synth.random_tree(cfg["n_otu"], cfg["seed"], 0.02, 0.15)
    P = cfg["n_pattern"] if n_pattern is None else int(n_pattern)
    st = synth.simulate_states(tree, P, cfg["ns"], cfg["seed"] if seed is None else seed, site_offset=pattern_offset)
    cfg["n_pattern"] = P
    return dict(tree=tree, states=st, model=model_block(cfg["model"]), cfg=cfg)


def algorithmic_bytes_per_pattern(n_otu, ns, ncatg):
    """bytes one full post-order Lk(NULL) must move per pattern with FP64 partials,
    int32 scalers and 1-byte tip states: (n-3) internal child reads + (n-2) writes of (C*S*8+4) bytes,
    (n-1) tip-state bytes, plus the root edge (C*S*8 + 4 + 1 + 8)."""
    rec = ncatg * ns * 8 + 4
    return (n_otu - 3) * rec + (n_otu - 2) * rec + (n_otu - 1) * 1 + (ncatg * ns * 8 + 4 + 1 + 8)
