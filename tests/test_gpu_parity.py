"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle and the reference's golden vectors.

Bars: partial vectors and scale vectors bit-equal to the oracle when fed the same
transition matrices; per-site log-likelihoods within 1e-10 absolute; lnL within 1e-10 relative of the
oracle and 1e-6 relative of the reference's AVX value (north-star tolerance; observed ~1e-15).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
from conftest import FIXTURES
from gpu_common import device_tree_from_golden


@pytest.fixture(scope="module")
def evaluated(golden):
    cache = {}

    def get(name):
        if name not in cache:
            d = golden(name)
            t, ot = device_tree_from_golden(d)
            t.Set_Both_Sides(True)
            lnl = t.Lk(None)
            ot_lnl = ot.lk(None, both_sides=True)
            cache[name] = (d, t, ot, lnl, ot_lnl, t.inst.site_outputs())
        return cache[name]
    yield get
    for v in cache.values():
        v[1].close()


@pytest.mark.parametrize("name", FIXTURES)
def test_lnl_matches_reference_and_oracle(name, evaluated):
    d, t, ot, lnl, ot_lnl, _ = evaluated(name)
    ref = d["lnL"][0]
    assert abs(lnl - ref) / abs(ref) < 1e-6          # north-star tolerance
    assert abs(lnl - ref) / abs(ref) < 1e-12         # what we actually hold
    assert abs(lnl - ot_lnl) / abs(ot_lnl) < 1e-12


@pytest.mark.parametrize("name", FIXTURES)
def test_site_outputs(name, evaluated):
    d, t, ot, lnl, _, (site, cur, cat, fact) = evaluated(name)
    w = d["wght"] > 0
    assert np.array_equal(fact[w], d["fact_sum_scale"][w])
    assert np.max(np.abs(site[w] - d["c_lnL_sorted"][w])) < 1e-10
    assert np.allclose(cat[w], d["unscaled_site_lk_cat"][w], rtol=1e-12, atol=0)
    assert np.allclose(cur[w], d["cur_site_lk"][w], rtol=1e-10, atol=0)


@pytest.mark.parametrize("name", FIXTURES)
def test_partials_and_scales_bit_equal(name, evaluated):
    d, t, ot, *_ = evaluated(name)
    w = d["wght"] > 0
    n = 0
    for (e, side), ref in ot.plk.items():
        got = t.partials(e, side)
        assert np.array_equal(got[w], ref[w]), (name, e, side)
        assert np.array_equal(t.scale_factors(e, side)[w], ot.scale[(e, side)][w]), (name, e, side)
        n += 1
    assert n == 3 * (t.n - 2)
    # and against the reference's own dump where it is held in full
    for e in d["full_edges"]:
        for side, nm in ((0, "left"), (1, "rght")):
            key = f"p_lk_{nm}_{e}"
            if key in d:
                assert np.array_equal(t.partials(int(e), side)[w], d[key][w])


@pytest.mark.parametrize("name", FIXTURES)
def test_lnl_at_every_edge(name, evaluated):
    d, t, ot, *_ = evaluated(name)
    got = np.array([t.Lk(e) for e in range(t.ne)])
    assert np.max(np.abs(got - d["edge_lnL"]) / np.abs(d["edge_lnL"])) < 1e-12


@pytest.mark.parametrize("name", FIXTURES)
def test_eigen_lr_and_dlk(name, evaluated):
    d, t, ot, *_ = evaluated(name)
    w = d["wght"] > 0
    for k, e in enumerate(d["eigen_edges"]):
        e = int(e)
        # Br_Len_Opt's call pattern (src/optimiz.c:622-630)
        t.Set_Update_Eigen_Lr(True); t.Set_Use_Eigen_Lr(False)
        t.Lk(e)
        t.Set_Update_Eigen_Lr(False); t.Set_Use_Eigen_Lr(True)
        dp = t.inst.get_dot_prod()
        ref = d[f"dot_prod_{e}"]
        assert np.allclose(dp[w], ref[w], rtol=1e-12, atol=1e-300)
        for j in range(3):
            l_in, lnl_ref, dlnl_ref = d["dlk_triples"][k, j]
            l_out, lnl = t.dLk(l_in, e)
            assert l_out == l_in
            assert abs(lnl - lnl_ref) / abs(lnl_ref) < 1e-12
            assert abs(t.c_dlnL - dlnl_ref) <= 1e-8 * max(1.0, abs(dlnl_ref))
        assert abs(t.Lk(e) - d[f"eig_lnL_{e}"][0]) / abs(d[f"eig_lnL_{e}"][0]) < 1e-12
        t.Set_Use_Eigen_Lr(False)


@pytest.mark.parametrize("name", ["nucleic_gtr_g4_inv", "synth_aa_90x24"])
def test_device_pmatrices(name, golden):
    """phyhip_update_transition_matrices (device PMat) vs the reference's P-matrices as dumped from the real PhyML: the same
    doubles (round 6: the device's exp() is the reference's libm's, phyml_amd/csrc/phyhip_exp.hpp; floor, renormalisation and the
    FMA chain of src/models.c:257-326 were the reference's already)."""
    d = golden(name)
    t, ot = device_tree_from_golden(d, host_pmat=False)
    try:
        lnl = t.Lk(None)
        npm = d["Pij_rr"].shape[0]
        for e in range(npm):
            got = t.inst.get_transition_matrix(e)
            assert np.array_equal(got, d["Pij_rr"][e]), e
        assert abs(lnl - d["lnL"][0]) / abs(d["lnL"][0]) < 1e-12
    finally:
        t.close()


def test_deferred_queue_is_transparent(golden):
    """One-op-at-a-time updates (SPR style) and a batched traversal give identical buffers."""
    d = golden("nucleic_gtr_g4")
    t1, ot = device_tree_from_golden(d)
    t2, _ = device_tree_from_golden(d)
    try:
        t1.Lk(None)
        for e in range(t2.ne):
            t2.Update_PMat_At_Given_Edge(e)
        order = []
        ot.post_order(ot.tip_root, ot.adj[ot.tip_root][0][0], ops=order)
        for (b, dd) in order:
            t2.Update_Partial_Lk(b, dd)
            t2.inst.synchronize()            # force a launch per operation
        b = ot.root_edge()
        l2 = t2.inst.edge_lnl(t2.side_buffer(b, 0), t2.side_buffer(b, 1), b)
        assert l2 == t1.c_lnL
        for (b, dd) in order:
            side = 0 if dd == ot.el[b] else 1
            assert np.array_equal(t1.partials(b, side), t2.partials(b, side))
    finally:
        t1.close(); t2.close()


def test_generic_loop_door(golden):
    """PHYHIP_FLAG_GENERIC_LOOP (the host layer sets it for mod->use_m4mod, `phyml --cov`): the device reproduces the reference's
    generic loop (Update_Partial_Lk_Generic, src/lk.c:1332-1587) -- the dumped partial vector and the scale vector bit for bit,
    every edge side bit-equal to the pinned restatement (arith = 2, tests/test_oracle_golden.py), lnL and per-site values."""
    d = golden("nucleic_cov_generic")
    t, ot = device_tree_from_golden(d, use_m4mod=True, arith=2)
    try:
        t.Set_Both_Sides(True)
        lnl = t.Lk(None)
        ref = ot.lk(None, both_sides=True)
        assert abs(lnl - d["lnL"][0]) / abs(d["lnL"][0]) < 1e-12 and abs(lnl - ref) / abs(ref) < 1e-12
        w = d["wght"] > 0
        assert np.array_equal(t.partials(0, 0)[w], np.asarray(d["p_lk_left_0"])[w])
        assert np.array_equal(t.scale_factors(0, 0)[w], np.asarray(d["sum_scale_left_0"])[w])
        for (e, side), p in ot.plk.items():
            # (every pattern: the generic loop writes zeros where the weight is zero, src/lk.c:1581-1584, and so does the device)
            assert np.array_equal(t.partials(e, side), p), (e, side)
            assert np.array_equal(t.scale_factors(e, side)[w], ot.scale[(e, side)][w]), (e, side)
        assert np.array_equal(t.partials(0, 0), np.asarray(d["p_lk_left_0"]))
        c_lnL_sorted, cur_site_lk, unscaled, fact = t.inst.site_outputs()
        assert np.array_equal(fact, d["fact_sum_scale"])
        assert np.max(np.abs(c_lnL_sorted[w] - d["c_lnL_sorted"][w]) / np.abs(d["c_lnL_sorted"][w])) < 1e-10
        # the SIMD-path instance of the same data differs from it exactly where the reference's two paths differ
        t2, _ = device_tree_from_golden(d)
        try:
            t2.Set_Both_Sides(True)
            t2.Lk(None)
            assert (t2.partials(0, 0)[w] != t.partials(0, 0)[w]).any(axis=1).sum() == 51
        finally:
            t2.close()
    finally:
        t.close()


def test_generic_loop_zero_weight_patterns():
    """The reference's generic loop zeroes the partial vector of a pattern without weight and leaves its scale exponent alone
    (src/lk.c:1405,1581-1584; the restatement with arith = 2 does the same): the device's generic-loop instances too -- every
    pattern of every buffer, not only the weighted ones."""
    from gpu_common import synthetic_pair
    rng = np.random.default_rng(4)
    wght = rng.integers(0, 3, 257).astype(np.float64)
    assert (wght == 0).sum() > 20
    t, ot, *_ = synthetic_pair(18, 257, 4, 4, seed=8, wght=wght, ambiguous_every=9, use_m4mod=True, arith=2)
    try:
        t.Set_Both_Sides(True)
        lnl, ref = t.Lk(None), ot.lk(None, both_sides=True)
        assert abs(lnl - ref) / abs(ref) < 1e-12
        for (e, side), p in ot.plk.items():
            got = t.partials(e, side)
            assert np.array_equal(got, p), (e, side)
            assert not got[wght == 0].any()
            assert np.array_equal(t.scale_factors(e, side), ot.scale[(e, side)]), (e, side)
    finally:
        t.close()
