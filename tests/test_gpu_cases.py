"""GPU parity beyond the golden fixtures: ragged and tiny pattern counts, every category-count instantiation,
ambiguity codes, scaling off, the compact-state tip API, error behaviour, and the BASELINE full-size configs
through size-independent properties (reference lnL of the regenerated input, pulley principle, shard additivity)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
import gpu_common
from gpu_common import synthetic_pair
from phyml_amd import capi, lktree, synth, workloads


def _check(t, ot, both=True, rel=1e-12):
    try:
        t.Set_Both_Sides(both)
        lnl = t.Lk(None)
        ref = ot.lk(None, both_sides=both)
        assert abs(lnl - ref) / abs(ref) < rel, (lnl, ref)
        w = ot.wght > 0
        n = 0
        for (e, side), p in ot.plk.items():
            if not both and not np.any(p):
                continue
            assert np.array_equal(t.partials(e, side)[w], p[w]), (e, side)
            assert np.array_equal(t.scale_factors(e, side)[w], ot.scale[(e, side)][w])
            n += 1
        assert n > 0
        site = t.inst.site_outputs()[0]
        assert np.max(np.abs(site[w] - ot.c_lnL_sorted[w])) < 1e-10
    finally:
        t.close()


@pytest.mark.parametrize("P", [1, 2, 15, 16, 17, 63, 64, 65, 255, 1000])
def test_ragged_pattern_counts_nt(P):
    t, ot, *_ = synthetic_pair(9, P, 4, 4, seed=20 + P, ambiguous_every=5)
    _check(t, ot)


@pytest.mark.parametrize("P", [1, 15, 16, 17, 33, 100])
def test_ragged_pattern_counts_aa(P):
    t, ot, *_ = synthetic_pair(8, P, 20, 4, seed=40 + P, ambiguous_every=6)
    _check(t, ot)


@pytest.mark.parametrize("ns,C", [(4, 1), (4, 2), (4, 3), (4, 5), (4, 8), (20, 1), (20, 2), (20, 3), (20, 5), (20, 8),
                                   (4, 9), (4, 12), (4, 16), (4, 33), (4, 64), (20, 12), (20, 16), (20, 40), (20, 64)])
def test_category_counts(ns, C):
    t, ot, *_ = synthetic_pair(10, 70, ns, C, seed=3 * C + ns, ambiguous_every=9)
    _check(t, ot)


@pytest.mark.parametrize("ns,C", [(4, 3), (4, 8), (20, 1), (20, 2), (20, 3), (20, 4), (20, 5), (20, 8), (4, 12), (4, 64), (20, 16), (20, 64)])
def test_device_built_matrices_at_every_category_count(ns, C):
    """phyhip_update_transition_matrices (src/lk.c:2344 -> src/models.c:257-326 on the device) against the restatement's
    matrices, BIT FOR BIT since round 6: the device computes exp() as the reference's libm does (phyml_amd/csrc/phyhip_exp.hpp),
    everything else was the reference's operation order already.  20 states: pmat20_kernel's A-operand table with the categories
    replicated over the four MFMA blocks (C = 1, 2), one block idle (C = 3), and -- more than four categories, the generic
    traversal kernel -- two passes per wave and no table; the whole tree's list and a short one (the arguments route)."""
    t, ot, *_ = synthetic_pair(10, 70, ns, C, seed=3 * C + ns, ambiguous_every=9, host_pmat=False)
    try:
        lnl = t.Lk(None)
        ref = ot.lk(None)
        assert abs(lnl - ref) / abs(ref) < 1e-12
        for e in range(ot.ne):
            assert np.array_equal(t.inst.get_transition_matrix(e), ot.pm[e]), e
        short = [1, 4, 6]
        lens = [0.031, 0.4, 2.2]
        t.inst.update_transition_matrices(np.array(short, np.int32), np.array(lens))
        for e, l in zip(short, lens):
            ot.len[e] = l
            ot.update_pmat(e)
            assert np.array_equal(t.inst.get_transition_matrix(e), ot.pm[e]), e
        # lengths over the whole range the clamp lets through (l_min ... l_max x rate: exp() of arguments down to the subnormal
        # results' branch), zero and negative lengths (src/lk.c:2296: MAX(0, l)), one matrix at a time and all at once
        rng = np.random.default_rng(11 * ns + C)
        lens = np.concatenate([10.0 ** rng.uniform(-9, 2.3, 3 * ot.ne - 3), [0.0, -0.25, 1e-300]])
        for k in range(0, len(lens), ot.ne):
            chunk = lens[k:k + ot.ne]
            idx = np.arange(len(chunk), dtype=np.int32)
            t.inst.update_transition_matrices(idx, chunk)
            for e, l in zip(idx, chunk):
                ot.len[e] = l
                ot.update_pmat(e)
                assert np.array_equal(t.inst.get_transition_matrix(int(e)), ot.pm[e]), (e, l)
    finally:
        t.close()


@pytest.mark.parametrize("ns,C", [(4, 12), (4, 16), (4, 40), (20, 12), (20, 16), (20, 40)])
def test_more_than_eight_categories_whole_surface(ns, C):
    """`phyml -c 12` is a legal command line (src/cl.c:1262-1263 takes any n >= 1).  Beyond 8 categories the plain lane =
    (pattern, category) kernels serve both state counts (a pattern's categories in up to 64 lanes of one wave), dLk's expl
    table no longer fits the kernel arguments and travels through device memory: the whole surface against the restatement --
    partials and scale vectors bit-equal on a tree deep enough to rescale, zero weights, ambiguity codes, Lk(b), Update_Eigen_Lr, dLk,
    the eigen-basis Lk."""
    n, P = (40, 150) if ns == 4 else (24, 70)
    w = np.ones(P); w[::7] = 0.0
    t, ot, tree, st = synthetic_pair(n, P, ns, C, seed=11 * C + ns, lmin=0.05, lmax=0.6, wght=w, ambiguous_every=8)
    try:
        t.Set_Both_Sides(True)
        lnl, ref = t.Lk(None), ot.lk(None, both_sides=True)
        assert abs(lnl - ref) / abs(ref) < 1e-12, (lnl, ref)
        ww = w > 0
        for (e, side), p in ot.plk.items():
            assert np.array_equal(t.partials(e, side)[ww], p[ww]), (e, side)
            assert np.array_equal(t.scale_factors(e, side)[ww], ot.scale[(e, side)][ww])
        assert np.array_equal(t.inst.site_outputs()[3][ww], ot.fact_sum_scale[ww])
        for e in (0, ot.ne // 2, ot.ne - 1):
            l0 = float(ot.len[e])
            assert abs(t.Lk(e) - ot.lk(e)) / abs(ref) < 1e-12
            t.Set_Update_Eigen_Lr(True); t.Set_Use_Eigen_Lr(False)
            t.Lk(e)
            ot.lk(e); ot.update_eigen_lr(e)
            t.Set_Update_Eigen_Lr(False); t.Set_Use_Eigen_Lr(True)
            assert np.allclose(t.inst.get_dot_prod()[ww], ot.dot_prod.reshape(P, -1)[ww], rtol=1e-12, atol=1e-300)
            for l in (0.5 * l0, l0, 3.0 * l0):
                l_out, v = t.dLk(l, e)
                lr, vr, dr = ot.dlk(l)
                assert l_out == lr and abs(v - vr) / abs(vr) < 1e-12 and abs(t.c_dlnL - dr) <= 1e-8 * max(1.0, abs(dr)), (e, l)
            assert abs(t.Lk(e) - ot.lk_eigen(l0)) / abs(ref) < 1e-12
            t.Set_Use_Eigen_Lr(False)
    finally:
        t.close()


@pytest.mark.parametrize("ns", [4, 20])
def test_scaling_switched_off_and_zero_weights(ns):
    w = np.ones(90); w[::4] = 0.0
    t, ot, *_ = synthetic_pair(12, 90, ns, 4, seed=5, wght=w, apply_scaling=0)
    _check(t, ot)


@pytest.mark.parametrize("ns", [4, 20])
def test_deep_tree_rescaling_and_post_order_only(ns):
    t, ot, *_ = synthetic_pair(260 if ns == 4 else 80, 48, ns, 4, seed=8, lmin=0.05, lmax=0.4)
    try:
        lnl = t.Lk(None)
        ref = ot.lk(None)
        assert abs(lnl - ref) / abs(ref) < 1e-12
        f = t.inst.site_outputs()[3]
        assert np.array_equal(f, ot.fact_sum_scale) and f.max() >= 256
    finally:
        t.close()


@pytest.mark.parametrize("ns", [4, 20])
def test_compact_tip_states_equal_tip_partials(ns):
    t, ot, tree, st = synthetic_pair(11, 130, ns, 4, seed=6)
    t2 = lktree.LkTree(11, tree.edge_left, tree.edge_rght, tree.edge_len, 130, ns, 4, host_pmat=True)
    try:
        m = ot.m
        t2.set_model(m.pi, m.gamma_rr, m.gamma_r_proba, m.e_val, m.r_e_vect, m.l_e_vect)
        t2.Make_Tree_For_Lk(np.ones(130))
        states = st.astype(np.int32).copy()
        states[0, :7] = ns + 3            # >= stateCount: fully ambiguous (BEAGLE compact-state convention)
        t2.set_tips(tip_states=states)
        tv = [v.copy() for v in ot.tip_vec]
        tv[0][:7, :] = 1.0
        t.set_tips(tip_partials=tv)
        assert t.Lk(None) == t2.Lk(None)
    finally:
        t.close(); t2.close()


def test_error_behaviour():
    inst = capi.Instance(4, 9, 4, 10, 5, 4)
    try:
        with pytest.raises(capi.PhyhipError):
            inst.set_tip_partials(0, np.full((10, 4), 0.5))          # not a 0/1 vector
        with pytest.raises(capi.PhyhipError):
            inst.update_partials([(2, 0, 0, 1, 1)])                    # destination is a tip
        with pytest.raises(capi.PhyhipError):
            inst.update_partials([(5, 0, 7, 1, 1)])                    # matrix index out of range
        with pytest.raises(capi.PhyhipError):
            inst.get_partials(99)
    finally:
        inst.close()
    with pytest.raises(capi.PhyhipError):
        capi.Instance(4, 9, 7, 10, 5, 4)                               # unsupported state count


def test_host_layer_exit_convention():
    """The C host layer prints and calls Exit() like the reference; the tests install a handler instead."""
    t, ot, *_ = synthetic_pair(6, 20, 4, 4, seed=2)
    try:
        with pytest.raises(capi.PhyhipError):
            t.dLk(float("nan"), 0)                                     # src/lk.c:671 assert(isnan(*l) == FALSE)
    finally:
        t.close()


def test_br_len_opt_increases_lnl():
    t, ot, *_ = synthetic_pair(14, 400, 4, 4, seed=12)
    try:
        t.Set_Both_Sides(True)
        l0 = t.Lk(None)
        e = 5
        t.edge(e).contents.l = t.edge(e).contents.l * 4.0             # spoil one branch length
        t.Update_PMat_At_Given_Edge(e)
        l1 = t.Lk(e)
        assert l1 < l0
        lopt, l2 = t.Br_Len_Newton(e)
        assert l2 >= l1 and l2 >= l0 - 1e-6 * abs(l0)
        assert abs(t.c_dlnL) < 1e-3 * abs(l2) * 1e-3 or abs(t.c_dlnL) < 1e-2
    finally:
        t.close()


@pytest.mark.parametrize("name,tol", [("cfg2_nt_100x50k", 1e-6), ("cfg3_aa_200x10k", 1e-6), ("small_nt_24x2000", 1e-6),
                                      ("small_aa_16x600", 1e-6)])
def test_baseline_configs_full_size(name, tol):
    """BASELINE configs at full size: the alignment is regenerated from its seed (checksum pinned), lnL must match
    the value the REAL reference's AVX path produced for it in the build container (north-star gate 1e-6 relative;
    we also hold 1e-12), be the same at several evaluation edges (pulley principle) and be additive over shards."""
    exp = workloads.manifest()["expected"][name]
    wl = workloads.make(name)
    tree, st, blk, cfg = wl["tree"], wl["states"], wl["model"], wl["cfg"]
    assert synth.states_checksum(st) == exp["checksum"]
    n, P, S, C = tree.n_otu, st.shape[1], cfg["ns"], int(blk["ncatg"][0])

    def build(lo, hi):
        t = lktree.LkTree(n, tree.edge_left, tree.edge_rght, tree.edge_len, hi - lo, S, C)
        t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"],
                    float(blk["l_min"][0]), float(blk["l_max"][0]))
        t.Make_Tree_For_Lk(np.ones(hi - lo))
        t.set_tips(tip_states=st[:, lo:hi].astype(np.int32))
        return t

    t = build(0, P)
    try:
        t.Set_Both_Sides(True)
        lnl = t.Lk(None)
        assert abs(lnl - exp["lnL"]) / abs(exp["lnL"]) < tol
        assert abs(lnl - exp["lnL"]) / abs(exp["lnL"]) < 1e-12
        for e in (0, 7, t.ne // 2, t.ne - 1):
            assert abs(t.Lk(e) - lnl) / abs(lnl) < 1e-11                   # src/lk.c:2642-2684
    finally:
        t.close()
    a, b = build(0, P // 3), build(P // 3, P)
    try:
        assert abs((a.Lk(None) + b.Lk(None)) - lnl) / abs(lnl) < 1e-12     # what the multi-GPU all-reduce relies on
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("name", ["nucleic_gtr_g4", "proteic_lg_g4"])
def test_tips_from_characters(name, golden):
    """Tips encoded by the host layer's character encoders (a14) give the reference's lnL on its own example alignments
    (IUPAC ambiguity, gaps)."""
    d = golden(name)
    t, ot = gpu_common.device_tree_from_golden(d)
    try:
        t.set_tips(tip_chars=d["tip_chars"])
        lnl = t.Lk()
        assert abs(lnl - float(d["lnL"][0])) <= 1e-12 * abs(lnl)
    finally:
        t.close()


def test_cfg4_shard_size_cross_checks_the_two_nucleotide_mappings():
    """125 000 patterns (one cfg4 shard) run on the one-lane-per-pattern mapping (G = 1), its halves on the two-lanes-per-
    pattern mapping (G = 2, chosen below ~82 k patterns): the shard lnLs must add up to the whole (1e-12) and the lnL must be
    the same at every evaluation edge (src/lk.c:2642-2684)."""
    wl = workloads.make("cfg4_nt_100x125k")
    tree, st, blk, cfg = wl["tree"], wl["states"], wl["model"], wl["cfg"]
    n, P = tree.n_otu, st.shape[1]
    assert P == 125000

    def build(lo, hi):
        t = lktree.LkTree(n, tree.edge_left, tree.edge_rght, tree.edge_len, hi - lo, 4, 4)
        t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"],
                    float(blk["l_min"][0]), float(blk["l_max"][0]))
        t.Make_Tree_For_Lk(np.ones(hi - lo))
        t.set_tips(tip_states=st[:, lo:hi].astype(np.int32))
        return t

    t = build(0, P)
    try:
        t.Set_Both_Sides(True)
        lnl = t.Lk(None)
        for e in (0, 11, t.ne - 1):
            assert abs(t.Lk(e) - lnl) / abs(lnl) < 1e-11
    finally:
        t.close()
    parts = 0.0
    for lo, hi in ((0, 60000), (60000, P)):
        s = build(lo, hi)
        try:
            parts += s.Lk(None)
        finally:
            s.close()
    assert abs(parts - lnl) / abs(lnl) < 1e-12, (parts, lnl)


@pytest.mark.parametrize("ns", [4, 20])
def test_model_setters_skip_unchanged_values_and_follow_changes(ns):
    """The model setters keep a host shadow: pushing the same block again (what update_beagle_* style callers do before
    every evaluation) must be a no-op, and any changed value must reach the device."""
    t, ot, *_ = synthetic_pair(9, 150, ns, 4, seed=31, host_pmat=False)
    try:
        m = ot.m
        a0 = t.Lk(None)
        assert abs(a0 - ot.lk(None)) <= 1e-12 * abs(a0)
        t.set_model(m.pi, m.gamma_rr, m.gamma_r_proba, m.e_val, m.r_e_vect, m.l_e_vect, m.l_min, m.l_max, 1.0, 1)
        assert t.Lk(None) == a0
        rr = m.gamma_rr[::-1].copy()                                   # same multiset of rates, other weights attached
        t.set_model(m.pi, rr, m.gamma_r_proba, m.e_val, m.r_e_vect, m.l_e_vect, m.l_min, m.l_max, 1.0, 1)
        b = t.Lk(None)
        assert b != a0
        pi2 = np.roll(m.pi, 1)
        t.set_model(pi2, rr, m.gamma_r_proba, m.e_val, m.r_e_vect, m.l_e_vect, m.l_min, m.l_max, 1.0, 1)
        assert t.Lk(None) != b
        t.set_model(m.pi, m.gamma_rr, m.gamma_r_proba, m.e_val, m.r_e_vect, m.l_e_vect, m.l_min, m.l_max, 2.0, 1)
        assert t.Lk(None) != a0                                        # br_len_mult travels with the options
        t.set_model(m.pi, m.gamma_rr, m.gamma_r_proba, m.e_val, m.r_e_vect, m.l_e_vect, m.l_min, m.l_max, 1.0, 1)
        assert t.Lk(None) == a0
    finally:
        t.close()


def test_update_lk_at_given_edge(golden):
    """Update_Lk_At_Given_Edge (src/lk.c:2478-2484): both sides of the edge recomputed from their neighbours, then Lk(b) --
    on an up-to-date tree it must return the tree's lnL at every edge, and it must repair a deliberately spoiled side."""
    d = golden("nucleic_gtr_g4")
    t, ot = gpu_common.device_tree_from_golden(d)
    try:
        t.Set_Both_Sides(True)
        lnl = t.Lk(None)
        for e in range(0, t.ne, 9):
            assert abs(t.Update_Lk_At_Given_Edge(e) - lnl) / abs(lnl) < 1e-12
        e = next(k for k in range(t.ne) if t.edge(k).contents.left.contents.tax == 0 and t.edge(k).contents.rght.contents.tax == 0)
        buf = t.side_buffer(e, 0)
        t.inst.set_partials(buf, np.full((t.P, t.C * t.S), 0.5))
        assert abs(t.Lk(e) - lnl) / abs(lnl) > 1e-6          # spoiled
        assert abs(t.Update_Lk_At_Given_Edge(e) - lnl) / abs(lnl) < 1e-12
    finally:
        t.close()


def test_alias_subpatt_gate_is_mirrored(golden):
    """`phyml --alias_subpatt` (src/lk.c:1294-1296): Update_Partial_Lk hands (the node opposite d, d) to the application's
    Alias_One_Subpatt before anything else -- for tips too, and not when the update flag of that side is off -- and the numbers
    do not change (no likelihood function reads that bookkeeping, include/phyhip_lk.h)."""
    import ctypes as C
    d = golden("nucleic_gtr_g4")
    t, ot = gpu_common.device_tree_from_golden(d)
    try:
        t.Set_Both_Sides(True)
        lnl = t.Lk(None)
        seen = []
        cb_t = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)
        num = lambda p: C.cast(p, C.POINTER(type(t.node(0).contents))).contents.num
        cb = cb_t(lambda a, dd, tree: seen.append((num(a), num(dd))))
        tr = t.tree.contents
        tr.alias_one_subpatt = C.cast(cb, C.c_void_p).value
        tr.do_alias_subpatt = 1
        tr.update_alias_subpatt = 0
        t.Lk(None)
        assert not seen                                        # both switches, like the reference
        tr.update_alias_subpatt = 1
        n0 = tr.n_edges_traversed
        assert t.Lk(None) == lnl
        # the traversals never reach Update_Partial_Lk with a tip (src/lk.c:285, 362): one call per partial update they make
        assert len(seen) == t.tree.contents.n_edges_traversed - n0 and len(seen) >= 2 * (t.n - 2)
        assert all(dd >= t.n for a, dd in seen)
        e = next(k for k in range(t.ne) if t.edge(k).contents.rght.contents.tax == 1)
        eb = t.edge(e).contents
        del seen[:]
        t.Update_Partial_Lk(e, eb.rght.contents.num)           # a tip: the call is made, nothing is queued
        assert seen == [(eb.left.contents.num, eb.rght.contents.num)]
        eb.update_partial_lk_rght = 0
        t.Update_Partial_Lk(e, eb.rght.contents.num)
        assert len(seen) == 1
        eb.update_partial_lk_rght = 1
    finally:
        t.close()


def test_rooted_input_tree_with_the_root_ignored(golden):
    """tree->n_root != NULL with ignore_root == YES (src/lk.c:420-429, 545-556, 573-576): the traversal starts from the two
    ends of the root edge and the likelihood is evaluated there; by the pulley principle it is the unrooted tree's value,
    on every choice of root edge, with one or both sides computed.  (The `phyml` program itself un-roots every input tree,
    src/main.c:215-216; ignore_root == NO cannot run on the reference's AVX path at all, oracle/probe_rooted.sh.)"""
    d = golden("nucleic_gtr_g4_inv")
    t, ot = gpu_common.device_tree_from_golden(d)
    try:
        ref = float(d["lnL"][0])
        for both in (False, True):
            t.Set_Both_Sides(both)
            for e in (0, 5, t.ne // 2, t.ne - 1):
                t.set_root_edge(e)
                assert abs(t.Lk(None) - ref) / abs(ref) < 1e-12, (both, e)
        t.set_root_edge(None)
        t.Set_Both_Sides(False)
        assert abs(t.Lk(None) - ref) / abs(ref) < 1e-12
    finally:
        t.close()


@pytest.mark.parametrize("ns,devices", [(4, None), (20, None), (4, [0, 0, 0]), (20, [0, 0])])
def test_single_character_tip_rewrite(ns, devices):
    """phyhip_set_tip_partials_at_pattern (Init_Partial_Lk_Tips_Double_One_Character, src/lk.c:2092: the tip rewrite of the
    leave-one-out cross-validation loops): hide one character, evaluate, restore it, evaluate -- against oracle trees built with
    the hidden / original character, partial vectors bit-equal; plain and sharded instances."""
    t, ot, tree, st = synthetic_pair(9, 130, ns, 4, seed=77, devices=devices)
    try:
        t.Set_Both_Sides(True)
        ref0 = ot.lk(None, both_sides=True)
        assert abs(t.Lk(None) - ref0) / abs(ref0) < 1e-12
        for tip, pat in ((3, 0), (0, 129), (8, 64)):
            rows = ot.tip_vec[tip].reshape(-1, ns)   # (a view: the oracle reads the same memory)
            orig = rows[pat].copy()
            hidden = np.ones(ns)
            t.inst.set_tip_partials_at_pattern(tip, pat, hidden)
            rows[pat] = hidden
            ot.tip_amb[tip][pat] = 1
            a, b = t.Lk(None), ot.lk(None, both_sides=True)
            assert abs(a - b) / abs(b) < 1e-12 and abs(b - ref0) > 1e-9
            for (e, side), p in ot.plk.items():
                assert np.array_equal(t.partials(e, side), p), (tip, pat, e, side)
            t.inst.set_tip_partials_at_pattern(tip, pat, orig)
            rows[pat] = orig
            ot.tip_amb[tip][pat] = 0
            a = t.Lk(None)
            assert abs(a - ref0) / abs(ref0) < 1e-12
    finally:
        t.close()
