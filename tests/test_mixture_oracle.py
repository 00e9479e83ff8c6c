"""Mixture path: the LG4X mixture of the reference's examples/lg4x (four class trees, each with
its own rate matrix, frequencies and rate multiplier; src/mixt.c) dumped from the REAL reference in XML mode
(oracle/mixt_driver.c).  The oracle evaluates every class tree as a plain single-category model and the restated site
loop of MIXT_Lk (phyml_amd/replay.py mixture_combine) must reproduce the reference's per-class site likelihoods,
scale exponents, per-site log-likelihoods and lnL."""
import os

import numpy as np
import pytest

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


# mixture_nt4: a four-class NUCLEOTIDE mixture (HKY85 + empirical frequencies / K80 + equal frequencies, four free rates) on
# examples/nucleic, dumped the same way (tests/golden/ntmix/nt4_check.xml, tests/golden/make_traces.py)
@pytest.mark.parametrize("fixture", ["mixture_lg4x", "mixture_nt4"])
def test_lg4x_mixture_matches_reference(fixture):
    d = phyg.load(os.path.join(GOLDEN, fixture + ".phyg"))
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


@pytest.mark.parametrize("fixture", ["mixture_lg4x_dlk", "mixture_nt4_dlk"])
def test_lg4x_mixture_dlk_matches_reference(fixture):
    """A MIXT_dLk call of the reference's own LG4X analysis (41st call of the first branch-length round): per-class
    eigen-basis products from the oracle's Update_Eigen_Lr on freshly computed partials, and the restated combination."""
    d = phyg.load(os.path.join(GOLDEN, fixture + ".phyg"))
    models, factors = replay.mixture_classes(d)
    e = int(d["eval_edge"][0])
    dots, facts = [], []
    for k, md in enumerate(models):
        ot = class_tree(d, md)
        ot.lk(both_sides=True)
        ot.update_eigen_lr(e)
        ref_dp = d[f"class{k}_dot_prod"]
        assert np.max(np.abs(ot.dot_prod - ref_dp) / np.maximum(np.abs(ref_dp), 1e-300)) < 1e-12
        left, rght = ot._side(e, 0), ot._side(e, 1)
        f = (ot.scale[(e, 0)] if (e, 0) in ot.scale else 0) + (ot.scale[(e, 1)] if (e, 1) in ot.scale else 0)
        assert np.array_equal(np.asarray(f, dtype=np.int64) + np.zeros(ot.P, dtype=np.int64), d[f"class{k}_fact"])
        dots.append(ot.dot_prod.copy()); facts.append(d[f"class{k}_fact"])
    args = (models, factors, float(d["r_mat_weight_sum"][0]), float(d["e_frq_weight_sum"][0]), float(d["sum_probas"][0]), d["wght"],
            float(d["dlk_l"][0]))
    lnl, dlnl = replay.mixture_dlk(dots, facts, *args)
    assert abs(lnl - float(d["lnL"][0])) < 1e-12 * abs(lnl)
    assert abs(dlnl - float(d["dlnL"][0])) < 1e-9 * max(1.0, abs(float(d["dlnL"][0])))
    lnl2, dlnl2 = replay.mixture_dlk([d[f"class{k}_dot_prod"] for k in range(4)], facts, *args)
    assert abs(lnl2 - float(d["lnL"][0])) < 1e-13 * abs(lnl2) and abs(dlnl2 - float(d["dlnL"][0])) < 1e-10 * max(1.0, abs(dlnl2))
