"""Mixture path: the LG4X mixture of the reference's examples/lg4x (four class trees, each with
its own rate matrix, frequencies and rate multiplier; src/mixt.c) dumped from the REAL reference in XML mode
(oracle/mixt_driver.c).  The oracle evaluates every class tree as a plain single-category model and the restated site
loop of MIXT_Lk (phyml_amd/replay.py mixture_combine) must reproduce the reference's per-class site likelihoods,
scale exponents, per-site log-likelihoods and lnL."""
import os

import numpy as np

from conftest import GOLDEN
from phyml_amd import phyg, replay
import orc


def class_tree(d, model_dict):
    m = orc.Model(model_dict)
    n = int(d["n_otu"][0])
    tv, ds, amb = replay.tips_from_masks(d["tip_mask"], m.ns)
    ot = orc.OracleTree(m, n, d["edge_left"], d["edge_rght"], d["edge_len"], d["wght"], tv, ds, amb, apply_scaling=1)
    ot.tip_root = 0  # MIXT_Lk evaluates at a_nodes[0]->b[0] (src/mixt.c:889)
    return ot


def test_lg4x_mixture_matches_reference():
    d = phyg.load(os.path.join(GOLDEN, "mixture_lg4x.phyg"))
    models, factors = replay.mixture_classes(d)
    assert len(models) == 4
    unscaled, fact = [], []
    for k, md in enumerate(models):
        ot = class_tree(d, md)
        e = int(d["eval_edge"][0])
        assert e == ot.root_edge()
        ot.lk()
        # transition matrices carry the class rate (src/lk.c:2298)
        assert np.array_equal(ot.pm[e][0], d[f"class{k}_Pij_eval_edge"])
        u = ot.unscaled_site_lk_cat[:, 0].copy(); f = ot.fact_sum_scale.copy()
        assert np.array_equal(f, d[f"class{k}_fact"])
        ref_u = d[f"class{k}_unscaled_site_lk_cat"]
        assert np.max(np.abs(u - ref_u) / ref_u) < 1e-13
        unscaled.append(u); fact.append(f)
    lnl, logs = replay.mixture_combine(unscaled, fact, factors, float(d["r_mat_weight_sum"][0]), float(d["e_frq_weight_sum"][0]),
                                       float(d["sum_probas"][0]), d["wght"])
    assert np.max(np.abs(logs - d["c_lnL_sorted"])) < 1e-11
    assert abs(lnl - float(d["lnL"][0])) < 1e-12 * abs(lnl)
    # and with the reference's own class likelihoods the restated combination is exact
    lnl2, logs2 = replay.mixture_combine([d[f"class{k}_unscaled_site_lk_cat"] for k in range(4)], [d[f"class{k}_fact"] for k in range(4)],
                                         factors, float(d["r_mat_weight_sum"][0]), float(d["e_frq_weight_sum"][0]),
                                         float(d["sum_probas"][0]), d["wght"])
    assert np.array_equal(logs2, d["c_lnL_sorted"]) and lnl2 == float(d["lnL"][0])
