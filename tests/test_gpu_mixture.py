"""GPU: the LG4X mixture (src/mixt.c) through the C ABI: four class instances (one category each, own
rate matrix / frequencies / class rate) + phyhip_calculate_mixture_log_likelihood, against the dump of the REAL
reference in XML mode (tests/golden/mixture_lg4x.phyg, oracle/mixt_driver.c)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from phyml_amd import capi, lktree, phyg, replay

pytestmark = pytest.mark.gpu


def _layouts():
    """None: plain instances.  Otherwise the class instances are SHARDED instances (pattern shards inside libphyhip.so, one RCCL
    all-reduce of {warning, lnL[, dlnL]} per mixture evaluation): one forced shard, three shards on device 0 and -- with two
    devices visible -- shards on distinct devices."""
    import torch
    lay = [("plain", None, False), ("one_shard_forced", [0], True), ("three_shards_dev0", [0, 0, 0], False)]
    if torch.cuda.device_count() >= 2:
        lay.append(("two_devices", [0, 1], False))
    return lay


LAYOUTS = _layouts()
LAY = pytest.mark.parametrize("layout", LAYOUTS, ids=[x[0] for x in LAYOUTS])
# lg4x: the reference's LG4X example (20 states); nt4: a four-class nucleotide mixture on examples/nucleic (HKY85 + empirical
# frequencies / K80 + equal frequencies, four free rates; tests/golden/ntmix/nt4_check.xml)
FIX = pytest.mark.parametrize("fx", ["lg4x", "nt4"])


@FIX
@LAY
@pytest.mark.parametrize("host_pmat", [True, False])
def test_lg4x_mixture_on_device(host_pmat, layout, fx):
    d = phyg.load(os.path.join(GOLDEN, f"mixture_{fx}.phyg"))
    models, factors = replay.mixture_classes(d)
    n, P, S = int(d["n_otu"][0]), int(d["n_pattern"][0]), int(d["ns"][0])
    tv, _, _ = replay.tips_from_masks(d["tip_mask"], S)
    trees = []
    try:
        for md in models:
            t = lktree.LkTree(n, d["edge_left"], d["edge_rght"], d["edge_len"], P, S, 1, host_pmat=host_pmat, devices=layout[1],
                              force_sharded=layout[2])
            t.tip_root = 0  # src/mixt.c:889
            t.set_model(md["pi"], md["gamma_rr"], md["gamma_r_proba"], md["e_val"], md["r_e_vect"], md["l_e_vect"],
                        float(md["l_min"][0]), float(md["l_max"][0]), float(md["br_len_mult"][0]), 1, 0, 0.0)
            t.Make_Tree_For_Lk(d["wght"], None)
            t.set_tips(tip_partials=tv)
            t.Lk()  # the class tree's own traversal (MIXT_Post_Order_Lk -> Update_Partial_Lk per class, src/mixt.c:1191-1250)
            trees.append(t)
        e = int(d["eval_edge"][0])
        ids = [t.tree.contents.b_inst for t in trees]
        parents = [t.side_buffer(e, 0) for t in trees]
        children = [t.side_buffer(e, 1) for t in trees]
        mats = [t.edge(e).contents.Pij_rr_idx for t in trees]
        lnl = capi.mixture_log_likelihood(ids, parents, children, mats, [f[0] for f in factors], [f[1] for f in factors],
                                          [f[2] for f in factors], float(d["r_mat_weight_sum"][0]),
                                          float(d["e_frq_weight_sum"][0]), float(d["sum_probas"][0]))
        ref = float(d["lnL"][0])
        assert abs(lnl - ref) <= (1e-13 if host_pmat else 1e-11) * abs(ref), (lnl, ref)
        logs = trees[0].inst.site_log_likelihoods()
        assert np.max(np.abs(logs - d["c_lnL_sorted"])) < (1e-11 if host_pmat else 1e-9)
        for k, t in enumerate(trees):  # per-class pieces the combination used
            _, _, u, f = t.inst.site_outputs()
            assert np.array_equal(np.asarray(f), d[f"class{k}_fact"])
            ref_u = d[f"class{k}_unscaled_site_lk_cat"]
            assert np.max(np.abs(np.asarray(u).reshape(-1) - ref_u) / ref_u) < (1e-13 if host_pmat else 1e-10)
    finally:
        for t in trees:
            t.close()


@FIX
@LAY
@pytest.mark.parametrize("host_pmat", [True, False])
def test_lg4x_mixture_dlk_on_device(host_pmat, layout, fx):
    """MIXT_dLk (src/mixt.c:2962-3340) through phyhip_calculate_mixture_eigen_lnl_dlnl against a call dumped from the
    reference's own LG4X analysis: class partials recomputed on both sides, Update_Eigen_Lr per class, combination."""
    d = phyg.load(os.path.join(GOLDEN, f"mixture_{fx}_dlk.phyg"))
    models, factors = replay.mixture_classes(d)
    n, P, S = int(d["n_otu"][0]), int(d["n_pattern"][0]), int(d["ns"][0])
    tv, _, _ = replay.tips_from_masks(d["tip_mask"], S)
    e = int(d["eval_edge"][0])
    trees = []
    try:
        for md in models:
            t = lktree.LkTree(n, d["edge_left"], d["edge_rght"], d["edge_len"], P, S, 1, host_pmat=host_pmat, devices=layout[1],
                              force_sharded=layout[2])
            t.tip_root = 0
            t.set_model(md["pi"], md["gamma_rr"], md["gamma_r_proba"], md["e_val"], md["r_e_vect"], md["l_e_vect"],
                        float(md["l_min"][0]), float(md["l_max"][0]), float(md["br_len_mult"][0]), 1, 0, 0.0)
            t.Make_Tree_For_Lk(d["wght"], None)
            t.set_tips(tip_partials=tv)
            t.Set_Both_Sides(1)
            t.Lk()
            t.Update_Eigen_Lr(e)   # MIXT_Update_Eigen_Lr: per class
            trees.append(t)
        ids = [t.tree.contents.b_inst for t in trees]
        lv, lnl, dlnl = capi.mixture_eigen_lnl_dlnl(ids, [t.side_buffer(e, 0) for t in trees], [t.side_buffer(e, 1) for t in trees],
                                                    float(d["dlk_l"][0]), [f[0] for f in factors], [f[1] for f in factors],
                                                    [f[2] for f in factors], float(d["r_mat_weight_sum"][0]),
                                                    float(d["e_frq_weight_sum"][0]), float(d["sum_probas"][0]))
        ref_lnl, ref_dlnl = float(d["lnL"][0]), float(d["dlnL"][0])
        tol = 1e-12 if host_pmat else 1e-10
        assert abs(lnl - ref_lnl) <= tol * abs(ref_lnl), (lnl, ref_lnl)
        assert abs(dlnl - ref_dlnl) <= 1e-8 * max(1.0, abs(ref_dlnl)), (dlnl, ref_dlnl)
    finally:
        for t in trees:
            t.close()


def _class_axis_tree(d, models, tv, host_matrices, layout=("plain", None, False)):
    """ONE instance whose four categories are the four LG4X classes (PHYHIP_FLAG_CLASS_AXIS): class rates as category
    rates, per-class frequencies / eigen systems pushed by index, matrices built on the device per class -- or computed per
    class by the oracle's PMat and uploaded (the bit-exact route)."""
    import orc
    n, P, S, K = int(d["n_otu"][0]), int(d["n_pattern"][0]), int(d["ns"][0]), len(models)
    t = lktree.LkTree(n, d["edge_left"], d["edge_rght"], d["edge_len"], P, S, K, host_pmat=False, class_axis=True, devices=layout[1],
                      force_sharded=layout[2])
    t.tip_root = 0
    m0 = models[0]
    rates = np.array([float(md["gamma_rr"][0]) for md in models])
    t.set_model(m0["pi"], rates, np.full(K, 1.0 / K), m0["e_val"], m0["r_e_vect"], m0["l_e_vect"], float(m0["l_min"][0]),
                float(m0["l_max"][0]), float(m0["br_len_mult"][0]), 1, 0, 0.0)
    t.Make_Tree_For_Lk(d["wght"], None)
    for k, md in enumerate(models):
        t.inst.set_state_frequencies(md["pi"], index=k)
        t.inst.set_eigen_decomposition(md["r_e_vect"], md["l_e_vect"], md["e_val"], index=k)
    t.set_tips(tip_partials=tv)
    if host_matrices:
        for e in range(t.ne):
            pm = np.concatenate([orc.pmat_edge(float(d["edge_len"][e]), S, 1, md["gamma_rr"], float(md["br_len_mult"][0]),
                                               float(md["l_min"][0]), float(md["l_max"][0]), md["r_e_vect"], md["l_e_vect"], md["e_val"])
                                 for md in models])
            t.inst.set_transition_matrix(t.edge(e).contents.Pij_rr_idx, pm)
    else:
        for e in range(t.ne):
            t.Update_PMat_At_Given_Edge(e)
    return t


@FIX
@LAY
@pytest.mark.parametrize("host_matrices", [True, False])
def test_lg4x_mixture_on_the_class_axis(host_matrices, layout, fx):
    """The same evaluation as test_lg4x_mixture_on_device with the four classes on the category axis of ONE instance:
    one traversal launch for all classes + the combination, against the reference's dump -- mixture lnL, per-site
    log-likelihoods, per-class likelihoods and per-class scale exponents."""
    d = phyg.load(os.path.join(GOLDEN, f"mixture_{fx}.phyg"))
    models, factors = replay.mixture_classes(d)
    S, K = int(d["ns"][0]), len(models)
    tv, _, _ = replay.tips_from_masks(d["tip_mask"], S)
    t = _class_axis_tree(d, models, tv, host_matrices, layout)
    try:
        root = t.node(0).contents.v[0].contents.num
        t.Post_Order_Lk(0, root)     # queues every update once; all classes run in the one launch below
        e = int(d["eval_edge"][0])
        lnl = t.inst.class_mixture_log_likelihood(t.side_buffer(e, 0), t.side_buffer(e, 1), t.edge(e).contents.Pij_rr_idx,
                                                  [f[0] for f in factors], [f[1] for f in factors], [f[2] for f in factors],
                                                  float(d["r_mat_weight_sum"][0]), float(d["e_frq_weight_sum"][0]),
                                                  float(d["sum_probas"][0]))
        ref = float(d["lnL"][0])
        assert abs(lnl - ref) <= (1e-13 if host_matrices else 1e-11) * abs(ref), (lnl, ref)
        logs = t.inst.site_log_likelihoods()
        assert np.max(np.abs(logs - d["c_lnL_sorted"])) < (1e-11 if host_matrices else 1e-9)
        _, _, u, f = t.inst.site_outputs(n_fact=K)
        u = np.asarray(u).reshape(-1, K); f = np.asarray(f).reshape(K, -1)
        for k in range(K):
            assert np.array_equal(f[k], d[f"class{k}_fact"])
            ref_u = d[f"class{k}_unscaled_site_lk_cat"]
            assert np.max(np.abs(u[:, k] - ref_u) / ref_u) < (1e-13 if host_matrices else 1e-10)
            # the class's own scale vectors of the two edge sides (phyhip_get_class_scale_factors) add up to its exponent
            sides = [t.side_buffer(e, 0), t.side_buffer(e, 1)]
            tot = sum(t.inst.get_class_scale_factors(b, k) for b in sides if b >= t.n)
            assert np.array_equal(tot, f[k])
    finally:
        t.close()


@FIX
@LAY
@pytest.mark.parametrize("host_matrices", [True, False])
def test_lg4x_mixture_dlk_on_the_class_axis(host_matrices, layout, fx):
    """MIXT_dLk on the class axis: both sides of every edge for all classes in two launches, per-class eigen products in
    one, the combination in one."""
    d = phyg.load(os.path.join(GOLDEN, f"mixture_{fx}_dlk.phyg"))
    models, factors = replay.mixture_classes(d)
    S = int(d["ns"][0])
    tv, _, _ = replay.tips_from_masks(d["tip_mask"], S)
    e = int(d["eval_edge"][0])
    t = _class_axis_tree(d, models, tv, host_matrices, layout)
    try:
        root = t.node(0).contents.v[0].contents.num
        t.Post_Order_Lk(0, root)
        t.Pre_Order_Lk(0, root)
        t.Update_Eigen_Lr(e)
        lv, lnl, dlnl = t.inst.class_mixture_eigen_lnl_dlnl(t.side_buffer(e, 0), t.side_buffer(e, 1), float(d["dlk_l"][0]),
                                                          [f[0] for f in factors], [f[1] for f in factors], [f[2] for f in factors],
                                                          float(d["r_mat_weight_sum"][0]), float(d["e_frq_weight_sum"][0]),
                                                          float(d["sum_probas"][0]))
        ref_lnl, ref_dlnl = float(d["lnL"][0]), float(d["dlnL"][0])
        tol = 1e-12 if host_matrices else 1e-10
        assert abs(lnl - ref_lnl) <= tol * abs(ref_lnl), (lnl, ref_lnl)
        assert abs(dlnl - ref_dlnl) <= 1e-8 * max(1.0, abs(ref_dlnl)), (dlnl, ref_dlnl)
    finally:
        t.close()


def _many_classes(d, K):
    """K synthetic class models from a dump's own classes (cycled, each with its own rate multiplier) and K weights that sum
    to one: a profile mixture of the C10-C60 kind has one class tree per profile (src/mixt.c:2603-2640)."""
    base, _ = replay.mixture_classes(d)
    models, factors = [], []
    rng = np.random.default_rng(7)
    w = rng.uniform(0.2, 1.0, K); w = w / w.sum()
    for k in range(K):
        md = dict(base[k % len(base)])
        md["gamma_rr"] = np.array([float(md["gamma_rr"][0]) * (0.35 + 0.11 * k)])
        models.append(md)
        factors.append((float(w[k]), 1.0, 1.0))
    return models, factors


@pytest.mark.parametrize("fx,K", [("nt4", 24), ("lg4x", 60)])
def test_many_class_mixture_against_the_oracle(fx, K):
    """More classes than the reference's example analyses have (phyhip_calculate_mixture_* take up to 64 class instances):
    MIXT_Lk and MIXT_dLk over K class instances against the oracle's per-class evaluation + the restated combinations
    (replay.mixture_combine / mixture_dlk, pinned to the reference's dumps by tests/test_mixture_oracle.py)."""
    from test_mixture_oracle import class_tree
    d = phyg.load(os.path.join(GOLDEN, f"mixture_{fx}_dlk.phyg"))
    models, factors = _many_classes(d, K)
    n, P, S = int(d["n_otu"][0]), int(d["n_pattern"][0]), int(d["ns"][0])
    tv, _, _ = replay.tips_from_masks(d["tip_mask"], S)
    e = int(d["eval_edge"][0])
    l = float(d["dlk_l"][0])
    # the oracle, class by class
    unscaled, fact, dots = [], [], []
    for md in models:
        ot = class_tree(d, md)
        ot.lk(both_sides=True)
        ot.lk(e)
        unscaled.append(ot.unscaled_site_lk_cat[:, 0].copy()); fact.append(ot.fact_sum_scale.copy())
        ot.update_eigen_lr(e)
        dots.append(ot.dot_prod.copy())
    ref_lnl, ref_logs = replay.mixture_combine(unscaled, fact, factors, float(K), float(K), 1.0, d["wght"])
    ref_l2, ref_dlnl = replay.mixture_dlk(dots, fact, models, factors, float(K), float(K), 1.0, d["wght"], l)
    trees = []
    try:
        for md in models:
            t = lktree.LkTree(n, d["edge_left"], d["edge_rght"], d["edge_len"], P, S, 1, host_pmat=True)
            t.tip_root = 0
            t.set_model(md["pi"], md["gamma_rr"], md["gamma_r_proba"], md["e_val"], md["r_e_vect"], md["l_e_vect"],
                        float(md["l_min"][0]), float(md["l_max"][0]), float(md["br_len_mult"][0]), 1, 0, 0.0)
            t.Make_Tree_For_Lk(d["wght"], None)
            t.set_tips(tip_partials=tv)
            t.Set_Both_Sides(1)
            t.Lk()
            trees.append(t)
        ids = [t.tree.contents.b_inst for t in trees]
        par = [t.side_buffer(e, 0) for t in trees]; chi = [t.side_buffer(e, 1) for t in trees]
        lnl = capi.mixture_log_likelihood(ids, par, chi, [t.edge(e).contents.Pij_rr_idx for t in trees], [f[0] for f in factors],
                                          [f[1] for f in factors], [f[2] for f in factors], float(K), float(K), 1.0)
        assert abs(lnl - ref_lnl) <= 1e-12 * abs(ref_lnl), (lnl, ref_lnl)
        assert np.max(np.abs(trees[0].inst.site_log_likelihoods() - ref_logs)) < 1e-10
        for t in trees:
            t.Update_Eigen_Lr(e)
        _, lnl2, dlnl = capi.mixture_eigen_lnl_dlnl(ids, par, chi, l, [f[0] for f in factors], [f[1] for f in factors],
                                                     [f[2] for f in factors], float(K), float(K), 1.0)
        assert abs(lnl2 - ref_l2) <= 1e-12 * abs(ref_l2), (lnl2, ref_l2)
        assert abs(dlnl - ref_dlnl) <= 1e-8 * max(1.0, abs(ref_dlnl)), (dlnl, ref_dlnl)
        # one class more than the tables hold is refused, not truncated
        if K == 60:
            with pytest.raises(Exception):
                capi.mixture_log_likelihood(ids + ids[:5], par + par[:5], chi + chi[:5],
                                            [t.edge(e).contents.Pij_rr_idx for t in trees] + [0] * 5, [0.0] * 65, [1.0] * 65,
                                            [1.0] * 65, float(K), float(K), 1.0)
    finally:
        for t in trees:
            t.close()


def _class_groups(K, S):
    """group sizes the class axis holds: up to 4 classes (20 states: 1-4; nucleotides: 1, 2 or 4)"""
    sizes, left = [], K
    while left > 0:
        g = min(4, left)
        if S == 4 and g == 3:
            g = 2
        sizes.append(g); left -= g
    return sizes


@pytest.mark.parametrize("fx,K", [("nt4", 3), ("nt4", 5), ("nt4", 10), ("nt4", 24), ("lg4x", 3), ("lg4x", 5), ("lg4x", 10), ("lg4x", 24),
                                  ("lg4x", 60)])
def test_many_class_mixture_in_groups_on_the_class_axis(fx, K):
    """Any class count on the class axis: the K classes ride in groups of up to four on the category axes of ceil(K / 4)
    instances (phyhip_calculate_mixture_* take class-axis instances as list entries) -- MIXT_Lk and MIXT_dLk against the oracle's
    per-class evaluation + the restated combinations, as for the one-instance-per-class form above."""
    from test_mixture_oracle import class_tree
    d = phyg.load(os.path.join(GOLDEN, f"mixture_{fx}_dlk.phyg"))
    models, factors = _many_classes(d, K)
    S = int(d["ns"][0])
    tv, _, _ = replay.tips_from_masks(d["tip_mask"], S)
    e = int(d["eval_edge"][0])
    l = float(d["dlk_l"][0])
    unscaled, fact, dots = [], [], []
    for md in models:
        ot = class_tree(d, md)
        ot.lk(both_sides=True)
        ot.lk(e)
        unscaled.append(ot.unscaled_site_lk_cat[:, 0].copy()); fact.append(ot.fact_sum_scale.copy())
        ot.update_eigen_lr(e)
        dots.append(ot.dot_prod.copy())
    ref_lnl, ref_logs = replay.mixture_combine(unscaled, fact, factors, float(K), float(K), 1.0, d["wght"])
    ref_l2, ref_dlnl = replay.mixture_dlk(dots, fact, models, factors, float(K), float(K), 1.0, d["wght"], l)
    sizes = _class_groups(K, S)
    assert sum(sizes) == K and len(sizes) == (K + 3) // 4 + (1 if (S == 4 and K % 4 == 3) else 0)
    trees, at = [], 0
    try:
        for g in sizes:
            t = _class_axis_tree(d, models[at:at + g], tv, True)
            at += g
            root = t.node(0).contents.v[0].contents.num
            t.Post_Order_Lk(0, root)
            t.Pre_Order_Lk(0, root)
            trees.append(t)
        ids = [t.tree.contents.b_inst for t in trees]
        par = [t.side_buffer(e, 0) for t in trees]; chi = [t.side_buffer(e, 1) for t in trees]
        pms = [t.edge(e).contents.Pij_rr_idx for t in trees]
        L = capi.load()
        import ctypes as C
        ia = lambda v: (C.c_int * len(v))(*[int(x) for x in v])
        da = lambda v: (C.c_double * len(v))(*[float(x) for x in v])
        out = C.c_double(0.0)
        rc = L.phyhip_calculate_mixture_log_likelihood(ia(ids), len(ids), ia(par), ia(chi), ia(pms), da([f[0] for f in factors]),
                                                       da([f[1] for f in factors]), da([f[2] for f in factors]), C.c_double(float(K)),
                                                       C.c_double(float(K)), C.c_double(1.0), C.byref(out))
        assert rc >= 0, L.phyhip_get_last_error()
        lnl = out.value
        assert abs(lnl - ref_lnl) <= 1e-12 * abs(ref_lnl), (lnl, ref_lnl)
        assert np.max(np.abs(trees[0].inst.site_log_likelihoods() - ref_logs)) < 1e-10
        for t in trees:
            t.Update_Eigen_Lr(e)
        lv, o1, o2 = C.c_double(l), C.c_double(0.0), C.c_double(0.0)
        rc = L.phyhip_calculate_mixture_eigen_lnl_dlnl(ia(ids), len(ids), ia(par), ia(chi), C.byref(lv), da([f[0] for f in factors]),
                                                       da([f[1] for f in factors]), da([f[2] for f in factors]), C.c_double(float(K)),
                                                       C.c_double(float(K)), C.c_double(1.0), C.byref(o1), C.byref(o2))
        assert rc >= 0, L.phyhip_get_last_error()
        assert abs(o1.value - ref_l2) <= 1e-12 * abs(ref_l2), (o1.value, ref_l2)
        assert abs(o2.value - ref_dlnl) <= 1e-8 * max(1.0, abs(ref_dlnl)), (o2.value, ref_dlnl)
    finally:
        for t in trees:
            t.close()
