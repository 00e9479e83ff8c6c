"""Virtual buffers (include/phyhip.h: phyhip_set_virtual_buffers): a whole-tree traversal does not store its tip x tip results,
every reader gets them recomputed in registers, and whatever LEAVES through the interface is still the double the reference
holds in t_edge::p_lk_* at that point -- including the values computed with matrices that have changed since.

Oracle: tests/orc.py (the pinned CPU restatement, host-computed matrices: bit-exact route) and the same device tree with the
feature switched off."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gpu_common import synthetic_pair

SHAPES = [(4, 4, 700), (4, 1, 300), (4, 2, 513), (4, 3, 200), (20, 4, 130), (20, 1, 90), (20, 2, 70)]


def cherry_keys(ot):
    """(edge, side) of every buffer whose two children are tips"""
    out = []
    for (e, side) in ot.plk:
        d = int(ot.el[e] if side == 0 else ot.er[e])
        kids = [v for (v, be) in ot.adj[d] if be != e]
        if all(v < ot.n for v in kids):
            out.append((e, side))
    return out


@pytest.mark.parametrize("ns,C,P", SHAPES)
@pytest.mark.parametrize("both", [False, True])
def test_full_traversal_virtual_buffers_read_back_bit_equal(ns, C, P, both):
    t, ot, tree, st = synthetic_pair(40, P, ns, C, seed=11 + ns + C, ambiguous_every=17)
    t0, _, _, _ = synthetic_pair(40, P, ns, C, seed=11 + ns + C, ambiguous_every=17)
    try:
        t0.inst.set_virtual_buffers(0)
        for x in (t, t0):
            x.Set_Both_Sides(both)
        lnl, lnl0 = t.Lk(None), t0.Lk(None)
        ref = ot.lk(None, both_sides=both)
        assert lnl == lnl0                      # the same doubles whether the stores were skipped or not
        assert abs(lnl - ref) / abs(ref) < 1e-12
        now, skipped, reissued, stored = t.inst.virtual_stats()
        n_cherry = len(cherry_keys(ot))
        assert now > 0 and skipped == now and stored == 0, (now, skipped, reissued, stored)
        assert now <= n_cherry
        assert t0.inst.virtual_stats() == (0, 0, 0, 0)
        # a second traversal of the same tree: again nothing stored for them
        assert t.Lk(None) == lnl
        assert t.inst.virtual_stats()[0] == now and t.inst.virtual_stats()[3] == 0
        # reading back materialises: every buffer (cherries included) bit-equal to the oracle and to the storing run
        keys = [k for k in ot.plk if both or np.any(ot.plk[k] != 0)]
        for k in keys:
            got, sc = t.partials(*k), t.scale_factors(*k)
            assert np.array_equal(got, ot.plk[k]), k
            assert np.array_equal(sc, ot.scale[k]), k
            assert np.array_equal(got, t0.partials(*k)), k
        assert t.inst.virtual_stats()[0] == 0 and t.inst.virtual_stats()[3] == now
        # and the evaluation at every edge afterwards (both sides of every edge are up to date only then)
        if both:
            assert t.Lk(None) == t0.Lk(None)   # (virtual again)
            for e in range(t.ne):
                assert t.Lk(e) == t0.Lk(e)
    finally:
        t.close(); t0.close()


@pytest.mark.parametrize("ns,C,P", [(4, 4, 333), (20, 4, 77)])
@pytest.mark.parametrize("host_pmat", [True, False])
def test_a_virtual_buffer_keeps_the_old_matrix_value(ns, C, P, host_pmat):
    """The reference's p_lk of a cherry stays what the last Update_Partial_Lk wrote, whatever happens to the pendant edges'
    matrices afterwards.  One matrix changed (stored first, or snapshot by the upload), and a whole-tree batch (snapshots by
    pmat_kernel)."""
    t, ot, tree, st = synthetic_pair(30, P, ns, C, seed=5, host_pmat=host_pmat)
    t0, _, _, _ = synthetic_pair(30, P, ns, C, seed=5, host_pmat=host_pmat)
    try:
        t0.inst.set_virtual_buffers(0)
        assert t.Lk(None) == t0.Lk(None)
        ck = cherry_keys(ot)
        n_virtual = t.inst.virtual_stats()[0]
        assert n_virtual > 0
        before = {k: (t0.partials(*k).copy(), t0.scale_factors(*k).copy()) for k in ck}
        # (1) one pendant edge of one cherry gets a new length and a new matrix -- nothing is recomputed
        e0, s0 = ck[0]
        d = int(ot.el[e0] if s0 == 0 else ot.er[e0])
        pend = [be for (v, be) in ot.adj[d] if be != e0][0]
        for x in (t, t0):
            x.edge(pend).contents.l = 0.37
            x.Update_PMat_At_Given_Edge(pend)
        for k in ck:
            assert np.array_equal(t.partials(*k), before[k][0]), k
            assert np.array_equal(t.scale_factors(*k), before[k][1]), k
        # (2) again virtual, then EVERY matrix changes in one batch (Update_All_PMat) without a traversal
        assert t.Lk(None) == t0.Lk(None)
        assert t.inst.virtual_stats()[0] == n_virtual
        before = {k: (t0.partials(*k).copy(), t0.scale_factors(*k).copy()) for k in ck}
        idx = np.array([t.edge(e).contents.Pij_rr_idx for e in range(t.ne)], dtype=np.int32)
        lens = np.linspace(0.01, 0.5, t.ne)
        if host_pmat:
            for e in range(t.ne):
                for x in (t, t0):
                    x.edge(e).contents.l = float(lens[e])
                    x.Update_PMat_At_Given_Edge(e)
        else:
            for x in (t, t0):
                x.inst.update_transition_matrices(idx, lens)
        assert t.inst.virtual_stats()[0] > 0    # snapshots, not stores
        for k in ck:
            assert np.array_equal(t.partials(*k), before[k][0]), k
            assert np.array_equal(t.scale_factors(*k), before[k][1]), k
        # the new matrices are what the next evaluation uses
        if not host_pmat:
            for x in (t, t0):
                for e in range(t.ne):
                    x.edge(e).contents.l = float(lens[e])
        assert t.Lk(None) == t0.Lk(None)
    finally:
        t.close(); t0.close()


@pytest.mark.parametrize("ns,C,P", [(4, 4, 450), (20, 4, 60)])
def test_search_like_stream_after_a_virtualising_traversal(ns, C, P):
    """Lk(NULL) leaves the cherries virtual; then the call pattern of a search: Lk(b) at every edge, partial updates along
    paths, Update_Eigen_Lr + dLk at edges whose sides are cherries, tip rewrites -- every scalar the storing run's."""
    t, ot, tree, st = synthetic_pair(36, P, ns, C, seed=3, host_pmat=False)
    t0, _, _, _ = synthetic_pair(36, P, ns, C, seed=3, host_pmat=False)
    try:
        t0.inst.set_virtual_buffers(0)
        for x in (t, t0):
            x.Set_Both_Sides(True)
        assert t.Lk(None) == t0.Lk(None)
        assert t.inst.virtual_stats()[0] > 0
        ck = cherry_keys(ot)
        rng = np.random.default_rng(1)
        for rep in range(3):
            for (e, side) in ck[:6]:
                d = int(ot.el[e] if side == 0 else ot.er[e])
                pend = [be for (v, be) in ot.adj[d] if be != e]
                for b in (e, pend[0], pend[1]):
                    got, ref = [], []
                    for x, out in ((t, got), (t0, ref)):
                        x.Set_Update_Eigen_Lr(True); x.Set_Use_Eigen_Lr(False)
                        out.append(x.Lk(b))
                        x.Set_Update_Eigen_Lr(False); x.Set_Use_Eigen_Lr(True)
                        for l in (0.05, 0.21):
                            out.append(x.dLk(l, b)[1]); out.append(x.c_dlnL)
                        x.Set_Use_Eigen_Lr(False)
                        x.edge(b).contents.l = float(0.03 + 0.1 * rep)
                        x.Update_PMat_At_Given_Edge(b)
                        out.append(x.Lk(b))
                    assert got == ref, (rep, e, b)
            # a full traversal in between makes them virtual again
            assert t.Lk(None) == t0.Lk(None)
            assert t.inst.virtual_stats()[0] > 0
            # one character of a cherry's tip hidden and restored (src/cv.c): stored on the old row first
            (e, side) = ck[rep % len(ck)]
            d = int(ot.el[e] if side == 0 else ot.er[e])
            tip = [v for (v, be) in ot.adj[d] if be != e][0]
            keep = t0.partials(e, side).copy()
            pat = int(rng.integers(0, P))
            for x in (t, t0):
                x.inst.set_tip_partials_at_pattern(tip, pat, np.ones(ns))
            assert np.array_equal(t.partials(e, side), keep)
            assert t.Lk(None) == t0.Lk(None)
    finally:
        t.close(); t0.close()


def test_short_lists_store_everything():
    """Below the threshold (16 operations by default) nothing is left virtual; the threshold is the instance's to set."""
    t, ot, tree, st = synthetic_pair(12, 200, 4, 4, seed=2)
    try:
        t.Lk(None)
        assert t.inst.virtual_stats() == (0, 0, 0, 0)
        t.inst.set_virtual_buffers(2)
        lnl = t.Lk(None)
        assert t.inst.virtual_stats()[0] > 0
        ref = ot.lk(None)
        assert abs(lnl - ref) / abs(ref) < 1e-12
        t.inst.set_virtual_buffers(0)
        assert t.inst.virtual_stats()[0] == 0
        for k in ot.plk:
            if np.any(ot.plk[k] != 0):
                assert np.array_equal(t.partials(*k), ot.plk[k]), k
    finally:
        t.close()


@pytest.mark.parametrize("ns,C,P,devices", [(20, 4, 90, None), (4, 4, 300, None), (20, 2, 50, None), (4, 4, 9000, None), (4, 4, 700, (0, 0)),
                                            (20, 4, 120, (0, 0, 0))])
@pytest.mark.parametrize("host_pmat", [True, False])
def test_random_call_streams_keep_the_stored_runs_scalars(ns, C, P, devices, host_pmat):
    """What a search does between two full traversals, at random: new lengths / matrices on random edges (pendant edges of cherries
    among them), partial updates over whole subtrees (long lists that READ buffers an earlier launch left virtual, and lists
    that re-virtualise them), evaluations at random edges, full traversals in between -- every scalar equal to the run that stores
    every buffer, and every buffer read back at the end."""
    # (9 000 patterns: two wave shapes and the large-grid resident evaluator; devices: pattern shards on device 0, every shard its own
    # virtual buffers)
    dev = None if devices is None else list(devices)
    t, ot, tree, st = synthetic_pair(34, P, ns, C, seed=21, host_pmat=host_pmat, ambiguous_every=13, devices=dev)
    t0, _, _, _ = synthetic_pair(34, P, ns, C, seed=21, host_pmat=host_pmat, ambiguous_every=13, devices=dev)
    try:
        t0.inst.set_virtual_buffers(0)
        for x in (t, t0):
            x.Set_Both_Sides(True)
        # (9 000 patterns: a list-form launch runs two wave shapes, a short one runs one -- other block sums, the same patterns: the two
        # runs may take different forms for the same call once one of them stores a virtual buffer in front of it, so 1e-13 there)
        same = (lambda a, b: a == b) if P < 8192 else (lambda a, b: abs(a - b) <= 1e-13 * abs(b))
        assert same(t.Lk(None), t0.Lk(None))
        # (docs/history/tools/gpu_fuzz_virtual.sh runs this test over more seeds and longer streams: PHYHIP_FUZZ_SEED / PHYHIP_FUZZ_ITERS)
        rng = np.random.default_rng(int(os.environ.get("PHYHIP_FUZZ_SEED", "17")))
        internal = [e for e in range(t.ne) if ot.el[e] >= ot.n and ot.er[e] >= ot.n]
        keys = list(ot.plk)
        for it in range(int(os.environ.get("PHYHIP_FUZZ_ITERS", "300"))):
            act = int(rng.integers(0, 10))
            if act == 0:
                assert same(t.Lk(None), t0.Lk(None)), it
            elif act in (1, 2):
                for e in rng.choice(t.ne, size=int(rng.integers(1, 6)), replace=False):
                    l = float(rng.uniform(0.005, 0.4))
                    for x in (t, t0):
                        x.edge(int(e)).contents.l = l
                        x.Update_PMat_At_Given_Edge(int(e))
            elif act == 3:
                e = int(rng.choice(internal))
                a, d = (int(ot.el[e]), int(ot.er[e])) if rng.integers(0, 2) else (int(ot.er[e]), int(ot.el[e]))
                for x in (t, t0):
                    x.Post_Order_Lk(a, d)      # the whole subtree behind d: a long list
                if it % 2:
                    # (the order of Lk(b) in a search, src/lk.c:513-528: partial updates queued, THEN the edge's matrix, then the evaluation)
                    l = float(rng.uniform(0.005, 0.4))
                    for x in (t, t0):
                        x.edge(e).contents.l = l
                        x.Update_PMat_At_Given_Edge(e)
                assert same(t.Lk(e), t0.Lk(e)), it
            elif act == 4:
                e = int(rng.integers(0, t.ne))
                if it % 3 == 0 and ot.el[e] >= ot.n:
                    # one partial update queued, a matrix set, then the evaluation (an SPR candidate's order, src/spr.c:643-646)
                    l = float(rng.uniform(0.005, 0.4))
                    for x in (t, t0):
                        x.Update_Partial_Lk(e, int(ot.el[e]))
                        x.edge(e).contents.l = l
                        x.Update_PMat_At_Given_Edge(e)
                assert same(t.Lk(e), t0.Lk(e)), it
            elif act == 5:
                e = int(rng.choice(internal))
                got, ref = [], []
                for x, out in ((t, got), (t0, ref)):
                    x.Set_Update_Eigen_Lr(True); x.Set_Use_Eigen_Lr(False)
                    out.append(x.Lk(e))
                    x.Set_Update_Eigen_Lr(False); x.Set_Use_Eigen_Lr(True)
                    out.append(x.dLk(0.11, e)[1]); out.append(x.c_dlnL)
                    x.Set_Use_Eigen_Lr(False)
                assert all(same(u, v) or abs(u - v) <= 1e-9 * max(1.0, abs(v)) for u, v in zip(got, ref)) if P >= 8192 else got == ref, it
            elif act in (6, 7):
                # single partial updates, left QUEUED (Br_Len_Opt / SPR rewrite one node at a time -- also nodes whose buffer an
                # earlier traversal left virtual: a short launch that stores it)
                for _ in range(int(rng.integers(1, 4))):
                    e = int(rng.integers(0, t.ne))
                    d = int(ot.el[e] if rng.integers(0, 2) else ot.er[e])
                    if d >= ot.n:
                        for x in (t, t0):
                            x.Update_Partial_Lk(e, d)
            elif act == 8:
                k = keys[int(rng.integers(0, len(keys)))]     # a host reader at a random moment
                assert np.array_equal(t.partials(*k), t0.partials(*k)), (it, k)
                assert np.array_equal(t.scale_factors(*k), t0.scale_factors(*k)), (it, k)
            elif act == 9 and it % 4 == 0:
                t.inst.set_virtual_buffers(int(rng.choice([0, 2, 16, 40])))   # (the threshold moves; 0 stores what is virtual)
            else:
                tip, pat = int(rng.integers(0, ot.n)), int(rng.integers(0, P))
                v = np.zeros(ns); v[int(rng.integers(0, ns))] = 1.0
                if rng.integers(0, 2):
                    v[:] = 1.0
                for x in (t, t0):
                    x.inst.set_tip_partials_at_pattern(tip, pat, v)
        assert same(t.Lk(None), t0.Lk(None))
        for k in ot.plk:
            assert np.array_equal(t.partials(*k), t0.partials(*k)), k
            assert np.array_equal(t.scale_factors(*k), t0.scale_factors(*k)), k
    finally:
        t.close(); t0.close()


@pytest.mark.parametrize("ns,C,P", [(20, 4, 70), (4, 4, 2600), (4, 4, 200)])
def test_readers_and_setters_behind_a_long_queue(ns, C, P):
    """A whole traversal QUEUED (nothing evaluated yet) and then something that reads a tip x tip buffer from memory, or changes
    what it is computed from: the launch in between must leave that buffer stored, on the old values -- `phyhip_get_partials`,
    `Update_Eigen_Lr` as a kernel of its own (20 states; nucleotides beyond the fused range), a pendant edge's matrix, a tip row."""
    t, ot, tree, st = synthetic_pair(30, P, ns, C, seed=9, host_pmat=True, ambiguous_every=11)
    t0, _, _, _ = synthetic_pair(30, P, ns, C, seed=9, host_pmat=True, ambiguous_every=11)
    try:
        t0.inst.set_virtual_buffers(0)
        ck = cherry_keys(ot)
        root = t.node(0).contents.v[0].contents.num

        def queue_all():
            for x in (t, t0):
                x.Post_Order_Lk(0, root); x.Pre_Order_Lk(0, root)
        for x in (t, t0):
            x.Set_Both_Sides(True)
        assert t.Lk(None) == t0.Lk(None)
        # (1) read a cherry right behind the queued traversal
        queue_all()
        e, side = ck[0]
        assert np.array_equal(t.partials(e, side), t0.partials(e, side))
        assert np.array_equal(t.scale_factors(e, side), t0.scale_factors(e, side))
        # (2) Update_Eigen_Lr + dLk at edges one of whose sides is a cherry, behind the queued traversal
        for (e, side) in ck[:4]:
            queue_all()
            got, ref = [], []
            for x, out in ((t, got), (t0, ref)):
                x.Set_Update_Eigen_Lr(True); x.Update_Eigen_Lr(e); x.Set_Update_Eigen_Lr(False)
                x.Set_Use_Eigen_Lr(True)
                out.append(x.dLk(0.07, e)[1]); out.append(x.c_dlnL); out.append(x.Lk(e))
                x.Set_Use_Eigen_Lr(False)
            assert got == ref, e
        # (3) a pendant edge's matrix changes behind the queued traversal: the cherry keeps the value of the queued update
        e, side = ck[1]
        d = int(ot.el[e] if side == 0 else ot.er[e])
        pend = [be for (v, be) in ot.adj[d] if be != e][0]
        queue_all()
        for x in (t, t0):
            x.edge(pend).contents.l = 0.29
            x.Update_PMat_At_Given_Edge(pend)
        assert np.array_equal(t.partials(e, side), t0.partials(e, side))
        # (4) a tip row changes behind the queued traversal
        tip = [v for (v, be) in ot.adj[d] if be != e][0]
        queue_all()
        for x in (t, t0):
            x.inst.set_tip_partials_at_pattern(tip, 3, np.ones(ns))
        assert np.array_equal(t.partials(e, side), t0.partials(e, side))
        assert t.Lk(None) == t0.Lk(None)
        for k in ot.plk:
            assert np.array_equal(t.partials(*k), t0.partials(*k)), k
    finally:
        t.close(); t0.close()


@pytest.mark.parametrize("ns,C,P", [(4, 4, 150), (20, 4, 40)])
def test_a_short_update_of_a_virtual_buffer_makes_it_real(ns, C, P):
    """A tip x tip buffer left virtual by a full traversal is rewritten by a SHORT launch (a new pendant length, then
    Update_Partial_Lk of that node alone: what Br_Len_Opt does at a cherry): the buffer is what that update stored -- its old
    definition is never stored over it later, whatever triggers the storing of the others."""
    t, ot, tree, st = synthetic_pair(26, P, ns, C, seed=6, host_pmat=True)
    t0, _, _, _ = synthetic_pair(26, P, ns, C, seed=6, host_pmat=True)
    try:
        t0.inst.set_virtual_buffers(0)
        for x in (t, t0):
            x.Set_Both_Sides(True)
        assert t.Lk(None) == t0.Lk(None)
        ck = cherry_keys(ot)
        assert t.inst.virtual_stats()[0] >= len(ck) - 1
        (e, side) = ck[0]
        d = int(ot.el[e] if side == 0 else ot.er[e])
        pend = [be for (v, be) in ot.adj[d] if be != e][0]
        for x in (t, t0):
            x.edge(pend).contents.l = 0.33
            x.Update_PMat_At_Given_Edge(pend)
            x.Update_Partial_Lk(e, d)          # one operation: stored
        # something else makes the remaining virtual buffers real (an evaluation at another cherry's edge)
        (e2, side2) = ck[1]
        assert t.Lk(e2) == t0.Lk(e2)
        assert t.inst.virtual_stats()[0] == 0
        assert np.array_equal(t.partials(e, side), t0.partials(e, side))
        for b in range(t.ne):
            assert t.Lk(b) == t0.Lk(b), b
    finally:
        t.close(); t0.close()
