"""The resident evaluator (phyhip_kernels.hpp: resident_dlk_kernel): chains of dLk / eigen-basis Lk calls on small nucleotide
alignments are served by workgroups that stay on the device.  It must be invisible in the numbers -- the same doubles as
the kernel-launch path, the oracle's values to 1e-12 -- and must actually be the path that ran."""
import time

import numpy as np
import pytest

import orc  # noqa: F401
from gpu_common import device_tree_from_golden, synthetic_pair

pytestmark = pytest.mark.gpu


def _chain(t, ot, edges, pause=0.0):
    """Br_Len_Opt's call pattern (src/optimiz.c:607-663) on each edge: Lk(b) with update_eigen_lr, dLk at several lengths,
    Lk(b) in the eigen basis; device values and the oracle's."""
    dev, ref = [], []
    for e in edges:
        t.Set_Update_Eigen_Lr(True); t.Set_Use_Eigen_Lr(False)
        dev.append(t.Lk(e))
        ot.lk(e); ot.update_eigen_lr(e); ref.append(ot.lk(e, refresh_pmat=False))
        t.Set_Update_Eigen_Lr(False); t.Set_Use_Eigen_Lr(True)
        for i in range(6):
            l = 0.003 * (i + 1) * (1 + e % 3)
            _, lnl = t.dLk(l, e)
            dev += [lnl, t.c_dlnL]
            _, rl, rd = ot.dlk(l)
            ref += [rl, rd]
            if pause and i == 2:
                time.sleep(pause)  # longer than the idle time: the workgroups have left and are launched again
        dev.append(t.Lk(e))
        ref.append(ot.lk_eigen(float(ot.len[e])))
        t.Set_Use_Eigen_Lr(False)
    return dev, ref


@pytest.mark.parametrize("patterns", [60, 382, 1500, 4000])
def test_resident_evaluator_matches_the_launch_path_and_the_oracle(patterns, monkeypatch):
    """382 patterns: every workgroup polls the host; 1500: workgroup 0 relays the commands through device memory; 4000: the
    large-grid evaluator."""
    vals = {}
    for res in ("0", "1"):
        monkeypatch.setenv("PHYHIP_RESIDENT", res)
        t, ot, *_ = synthetic_pair(14, patterns, 4, 4, seed=23, ambiguous_every=17)
        try:
            t.Set_Both_Sides(True)
            t.Lk(None)
            ot.lk(None, both_sides=True)
            dev, ref = _chain(t, ot, [3, 11, 20, 3])
            # (4000 patterns = 125 tiles: beyond the small evaluators' 64 -- served by the large-grid one, phyhip_big.hpp)
            served, launches, silent, busy = t.inst.resident_stats(0 if patterns <= 2000 else 2)
            if res == "1":
                assert served >= 4 * 7 - (4 if patterns <= 2000 else 8) and silent == 0, (served, launches, silent, busy)
                assert launches <= 5  # (the oracle runs between the chains: the workgroups may have left meanwhile)
                if patterns <= 1500:
                    # the chain's Lk(b) and its Update_Eigen_Lr (partial update + eigen products: ONE command) are served by
                    # the short-launch evaluator (up to 64 one-wave workgroups: 2 048 patterns)
                    s_served, _, s_silent, _ = t.inst.resident_stats(1)
                    assert s_served >= 4 and s_silent == 0, (s_served, s_silent)
            else:
                assert served == 0 and launches == 0
            for a, b in zip(dev, ref):
                assert abs(a - b) <= 1e-12 * max(1.0, abs(b)), (a, b)
            assert t.inst.numerical_warning() == 0
            vals[res] = dev
        finally:
            t.close()
    assert vals["0"] == vals["1"]  # the same doubles, whichever way they were computed


def test_resident_workgroups_leave_when_idle_and_come_back(monkeypatch):
    monkeypatch.setenv("PHYHIP_RESIDENT_IDLE_US", "200")
    t, ot, *_ = synthetic_pair(10, 300, 4, 4, seed=4)
    try:
        t.Set_Both_Sides(True)
        t.Lk(None)
        ot.lk(None, both_sides=True)
        dev, ref = _chain(t, ot, [2, 9, 5], pause=0.003)
        served, launches, silent, busy = t.inst.resident_stats()
        assert served > 0 and launches >= 3 and silent == 0
        for a, b in zip(dev, ref):
            assert abs(a - b) <= 1e-12 * max(1.0, abs(b)), (a, b)
    finally:
        t.close()


def test_resident_evaluator_sees_what_the_stream_wrote_in_between(golden):
    """Weights, the invariant-site model and branch lengths change between chains (uploads and kernels on the instance's
    stream, which the resident workgroups are not ordered with): every chain still reproduces the golden triples."""
    d = golden("nucleic_gtr_g4_inv")
    t, ot = device_tree_from_golden(d)
    try:
        t.Set_Both_Sides(True)
        lnl0 = t.Lk(None)
        assert abs(lnl0 - d["lnL"][0]) / abs(d["lnL"][0]) < 1e-12
        for rep in range(3):
            for k, e in enumerate(d["eigen_edges"]):
                e = int(e)
                t.Set_Update_Eigen_Lr(True); t.Set_Use_Eigen_Lr(False)
                t.Lk(e)
                t.Set_Update_Eigen_Lr(False); t.Set_Use_Eigen_Lr(True)
                for j in range(3):
                    l_in, lnl_ref, dlnl_ref = d["dlk_triples"][k, j]
                    _, lnl = t.dLk(l_in, e)
                    assert abs(lnl - lnl_ref) / abs(lnl_ref) < 1e-12
                    assert abs(t.c_dlnL - dlnl_ref) <= 1e-8 * max(1.0, abs(dlnl_ref))
                t.Set_Use_Eigen_Lr(False)
            t.Lk(None)  # a full traversal on the stream between the chains
        served, launches, silent, busy = t.inst.resident_stats()
        assert served > 0 and silent == 0
    finally:
        t.close()


def test_two_instances_each_with_resident_workgroups():
    pairs = [synthetic_pair(9, 200 + 90 * k, 4, 4, seed=40 + k) for k in range(2)]
    try:
        for t, ot, *_ in pairs:
            t.Set_Both_Sides(True); t.Lk(None); ot.lk(None, both_sides=True)
        for rep in range(2):
            for t, ot, *_ in pairs:
                dev, ref = _chain(t, ot, [1 + rep, 6])
                for a, b in zip(dev, ref):
                    assert abs(a - b) <= 1e-12 * max(1.0, abs(b)), (a, b)
        for t, *_ in pairs:
            assert t.inst.resident_stats()[0] > 0
    finally:
        for t, *_ in pairs:
            t.close()


@pytest.mark.parametrize("taxa,patterns,categories", [(40, 382, 4), (30, 1500, 4), (24, 1200, 2), (20, 700, 1)])
def test_resident_short_evaluations_spr_call_pattern(taxa, patterns, categories, monkeypatch):
    """A seeded SPR / Br_Len_Opt call stream (phyml_amd/replay.py: matrix refreshes, partial updates, edge likelihoods, eigen
    products, dLk chains) with both resident evaluators against a kernel launch per call: the same doubles, scalar by
    scalar, and the oracle's values; device-built matrices (the resident workgroups rebuild them in their prologue)."""
    from phyml_amd import replay
    import replay_oracle
    res, stats = {}, {}
    for r in ("0", "1"):
        monkeypatch.setenv("PHYHIP_RESIDENT", r)
        t, ot, tree, st = synthetic_pair(taxa, patterns, 4, categories, seed=61, host_pmat=False, ambiguous_every=13)
        try:
            t.Set_Both_Sides(True)
            t.Lk(None)
            tr = replay.make_trace(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, 80, seed=8, walk_every=3, opt_every=4, n_dlk=4)
            res[r] = t.Replay_Surface_Trace(tr)
            stats[r] = (t.inst.resident_stats(0), t.inst.resident_stats(1))
            if r == "1":
                ot.lk(None, both_sides=True)
                ref, ref2 = replay_oracle.OracleReplayer(ot).run(tr)
                m = ref != 0
                assert np.max(np.abs(res[r][0][m] - ref[m]) / np.abs(ref[m])) < 1e-9  # (device exp in the matrices)
        finally:
            t.close()
    assert np.array_equal(res["0"][0], res["1"][0]) and np.array_equal(res["0"][1], res["1"][1])
    assert stats["0"] == ((0, 0, 0, 0), (0, 0, 0, 0))
    (d_served, _, d_silent, _), (t_served, t_launches, t_silent, _) = stats["1"]
    assert d_served > 0 and t_served > 50 and d_silent == 0 and t_silent == 0 and t_launches <= 6


def test_resident_protocol_under_stress(monkeypatch):
    """Idle time far below the time between chains: the workgroups leave and are launched again all the time, commands
    arrive while they are leaving (unanswered: detected through workgroup 0's exit report, the evaluation is then launched)
    -- and every scalar is still the launch path's double."""
    from phyml_amd import replay
    res = {}
    # (3 us is below the time between any two commands on any host: the workgroups leave after nearly every command; 15 us is
    # between the commands of a chain and the pause of an evaluator while the other one works -- on a slow enough host)
    for r, idle in (("0", "1000"), ("1", "3"), ("1", "15")):
        monkeypatch.setenv("PHYHIP_RESIDENT", r)
        monkeypatch.setenv("PHYHIP_RESIDENT_IDLE_US", idle)
        t, ot, tree, st = synthetic_pair(30, 500, 4, 4, seed=71, host_pmat=False)
        try:
            t.Set_Both_Sides(True)
            t.Lk(None)
            tr = replay.make_trace(30, tree.edge_left, tree.edge_rght, tree.edge_len, 300, seed=12, walk_every=3, opt_every=3, n_dlk=4)
            res[(r, idle)] = t.Replay_Surface_Trace(tr)
            if r == "1":
                d, s = t.inst.resident_stats(0), t.inst.resident_stats(1)
                assert d[0] + d[2] > 0 and s[0] + s[2] > 0 and d[1] + s[1] > (4 if idle == "3" else 1), (d, s)
        finally:
            t.close()
    a, a2 = res[("0", "1000")]
    for k, (b, b2) in res.items():
        assert np.array_equal(a, b) and np.array_equal(a2, b2), k


@pytest.mark.parametrize("taxa,patterns,categories", [(40, 382, 4), (30, 1500, 4), (20, 700, 1), (16, 5000, 4), (12, 40000, 4), (12, 7000, 3)])
def test_host_computed_matrices_ride_in_resident_commands(taxa, patterns, categories, monkeypatch):
    """The bit-exact matrix route -- the host layer's PMat() + phyhip_set_transition_matrix (src/lk.c:2360), what PhyML's glue uses by
    default -- in the search's call pattern: the three matrices of an SPR candidate travel to ResidentCtl::up_area through the BAR
    in front of the command, the resident workgroups (short-launch evaluator up to 2 048 patterns, large-grid evaluator beyond) take
    them to their slots as a launch takes them from its arguments.  The launch path's doubles, scalar by scalar; the oracle's to
    1e-12 (host matrices: no device exp); the candidates are served, not launched; the workgroups leaving all the time (commands
    that go unanswered are launched with their matrices queued again) changes nothing."""
    from phyml_amd import replay
    import replay_oracle
    big = patterns > 2048
    res, stats = {}, {}
    for r, idle in (("0", "1000"), ("1", "1000"), ("1", "5")):
        monkeypatch.setenv("PHYHIP_RESIDENT", r)
        monkeypatch.setenv("PHYHIP_RESIDENT_IDLE_US", idle)
        t, ot, tree, st = synthetic_pair(taxa, patterns, 4, categories, seed=63, host_pmat=True, ambiguous_every=13)
        try:
            t.Set_Both_Sides(True)
            t.Lk(None)
            tr = replay.make_trace(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, 80, seed=8, walk_every=3, opt_every=4, n_dlk=4)
            res[(r, idle)] = t.Replay_Surface_Trace(tr)
            stats[(r, idle)] = t.inst.resident_stats(2 if big else 1)
            if (r, idle) == ("1", "1000"):
                ot.lk(None, both_sides=True)
                ref, ref2 = replay_oracle.OracleReplayer(ot).run(tr)
                m = ref != 0
                assert np.max(np.abs(res[(r, idle)][0][m] - ref[m]) / np.abs(ref[m])) < 1e-12
        finally:
            t.close()
    a, a2 = res[("0", "1000")]
    for k, (b, b2) in res.items():
        assert np.array_equal(a, b) and np.array_equal(a2, b2), k
    served, launches, silent, busy = stats[("1", "1000")]
    assert stats[("0", "1000")][0] == 0
    # (needs the host's stores into device memory -- a large BAR, as on every MI355X box of this pool; without it such commands are launched)
    # every candidate's Lk(b) (+ the eigen products of the branch-length chains); a command that meets the workgroups as they leave
    # (the first dLk chain launches the other evaluator: milliseconds on a cold process) goes unanswered and is launched -- rare
    assert served >= 80 and silent <= 2, stats
    s2 = stats[("1", "5")]
    assert s2[0] + s2[2] > 0 and s2[1] > 1, s2


# ---- the large-grid resident evaluator (phyml_amd/csrc/phyhip_big.hpp) ----------------------------------------------------

@pytest.mark.parametrize("taxa,patterns,categories", [(16, 5000, 4), (12, 40000, 4), (12, 9000, 2), (12, 7000, 3), (12, 6000, 1),
                                                      (8, 140000, 4), (12, 7200, 4), (10, 20000, 3)])
def test_large_grid_resident_evaluator_spr_and_brlen_call_pattern(taxa, patterns, categories, monkeypatch):
    """Beyond 64 pattern tiles the scalar-returning calls of a search are served by resident_big_kernel -- one persistent
    workgroup per compute unit, commands relayed through device memory, a wave walks several tiles, the tile sums added by the
    host (up to 200 tiles: 5 000 patterns), on the device through tickets (fewer than 256 workgroups: 7 200 patterns = 225 tiles)
    or on the device through one partial sum per workgroup (256 workgroups: the other sizes) -- the seeded SPR / Br_Len_Opt stream
    returns the launch path's doubles, scalar by scalar, and the oracle's values.  Covers every instantiation: two lanes per pattern (4 and 2 categories, eight waves per
    workgroup), one lane per pattern (3 and 1 categories; 4 categories beyond 131 072 patterns)."""
    from phyml_amd import replay
    import replay_oracle
    res, stats = {}, {}
    ncand = 60 if patterns <= 40000 else 24
    for r in ("0", "1"):
        monkeypatch.setenv("PHYHIP_RESIDENT", r)
        t, ot, tree, st = synthetic_pair(taxa, patterns, 4, categories, seed=67, host_pmat=False, ambiguous_every=13)
        try:
            t.Set_Both_Sides(True)
            t.Lk(None)
            tr = replay.make_trace(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, ncand, seed=9, walk_every=3, opt_every=4, n_dlk=4)
            res[r] = t.Replay_Surface_Trace(tr)
            again = t.Lk(None)  # a long traversal launch after the resident phase: the workgroups make room
            res[r] = (res[r][0], res[r][1], again)
            stats[r] = t.inst.resident_stats(2)
            assert t.inst.numerical_warning() == 0
            if r == "1":
                full = ot.lk(None, both_sides=True)
                ref, ref2 = replay_oracle.OracleReplayer(ot).run(tr)
                m = ref != 0
                assert np.max(np.abs(res[r][0][m] - ref[m]) / np.abs(ref[m])) < 1e-9  # (device exp in the matrices)
                dl = tr["kind"] == replay.DLK
                assert np.max(np.abs(res[r][1][dl] - ref2[dl]) / np.maximum(1.0, np.abs(ref2[dl]))) < 1e-7
                assert abs(again - full) / abs(full) < 1e-11
        finally:
            t.close()
    assert np.array_equal(res["0"][0], res["1"][0]) and np.array_equal(res["0"][1], res["1"][1]) and res["0"][2] == res["1"][2]
    assert stats["0"] == (0, 0, 0, 0)
    served, launches, silent, busy = stats["1"]
    n_scalar = int(np.isin(tr["kind"], (replay.EDGE_LNL, replay.DLK, replay.EIGEN_LR)).sum())
    # (the both-sides traversal in front of the stream leaves the tip x tip buffers virtual, include/phyhip.h: the first evaluation
    # that reads one stores them all -- one launch instead of a resident command)
    assert served >= n_scalar - 9 and silent == 0 and launches <= 3, (stats["1"], n_scalar)


def test_large_grid_resident_evaluator_brlen_chains_weights_and_invariant_sites(monkeypatch):
    """Br_Len_Opt's chains (Lk(b) with Update_Eigen_Lr, dLk at several lengths, Lk(b) in the eigen basis) on 20 000 weighted
    patterns with the invariant-site model: resident against launched (the same doubles) against the oracle."""
    vals = {}
    rng = np.random.default_rng(5)
    P = 20000
    wght = rng.integers(0, 4, P).astype(np.float64)
    for res in ("0", "1"):
        monkeypatch.setenv("PHYHIP_RESIDENT", res)
        t, ot, *_ = synthetic_pair(14, P, 4, 4, seed=29, ambiguous_every=17, wght=wght)
        try:
            t.Set_Both_Sides(True)
            t.Lk(None)
            ot.lk(None, both_sides=True)
            dev, ref = _chain(t, ot, [3, 11, 20, 3, 7])
            for a, b in zip(dev, ref):
                assert abs(a - b) <= 1e-11 * max(1.0, abs(b)), (a, b)
            if res == "1":
                served, launches, silent, busy = t.inst.resident_stats(2)
                assert served >= 5 * 7 - 6 and silent == 0, (served, launches, silent, busy)
            vals[res] = dev
        finally:
            t.close()
    assert vals["0"] == vals["1"]


def test_large_grid_resident_workgroups_leave_and_come_back(monkeypatch):
    """Idle time below the time between commands: the workgroups leave all the time, commands meet nobody (detected,
    launched instead), long launches in between -- every scalar still the launch path's."""
    from phyml_amd import replay
    res = {}
    for r, idle in (("0", "1000"), ("1", "5"), ("1", "40")):
        monkeypatch.setenv("PHYHIP_RESIDENT", r)
        monkeypatch.setenv("PHYHIP_RESIDENT_IDLE_US", idle)
        t, ot, tree, st = synthetic_pair(20, 12000, 4, 4, seed=73, host_pmat=False)
        try:
            t.Set_Both_Sides(True)
            t.Lk(None)
            out = []
            for rep in range(3):
                tr = replay.make_trace(20, tree.edge_left, tree.edge_rght, tree.edge_len, 50, seed=12 + rep, walk_every=3, opt_every=3, n_dlk=3)
                out.append(t.Replay_Surface_Trace(tr))
                out.append((t.Lk(None),))
            res[(r, idle)] = out
            if r == "1":
                s = t.inst.resident_stats(2)
                assert s[0] + s[2] > 0 and s[1] >= 3, s
        finally:
            t.close()
    a = res[("0", "1000")]
    for k, b in res.items():
        for x, y in zip(a, b):
            assert all(np.array_equal(u, v) for u, v in zip(x, y)), k


def test_two_large_instances_share_one_device(monkeypatch):
    """Only one instance per device holds large-grid resident workgroups; the other one's evaluations are launched -- both
    return the oracle's values, whichever gets them."""
    from phyml_amd import replay
    import replay_oracle
    pairs = [synthetic_pair(10, 6000 + 3000 * k, 4, 4, seed=50 + k, host_pmat=False) for k in range(2)]
    try:
        for t, ot, tree, st in pairs:
            t.Set_Both_Sides(True); t.Lk(None); ot.lk(None, both_sides=True)
        for rep in range(2):
            for t, ot, tree, st in pairs:
                tr = replay.make_trace(10, tree.edge_left, tree.edge_rght, tree.edge_len, 20, seed=3 + rep, walk_every=3, opt_every=4, n_dlk=3)
                got, got2 = t.Replay_Surface_Trace(tr)
                ref, ref2 = replay_oracle.OracleReplayer(ot).run(tr)
                m = ref != 0
                assert np.max(np.abs(got[m] - ref[m]) / np.abs(ref[m])) < 1e-9
        assert sum(t.inst.resident_stats(2)[0] for t, *_ in pairs) > 0
    finally:
        for t, *_ in pairs:
            t.close()


@pytest.mark.parametrize("P,C", [(60, 4), (429, 4), (1000, 4), (429, 2), (200, 3), (300, 1)])
@pytest.mark.parametrize("host_pmat", [False, True])
def test_20_state_resident_evaluator_is_the_launch_path_bit_for_bit(P, C, host_pmat, monkeypatch):
    """Small 20-state alignments: SPR regraft candidates (src/spr.c:640-646: three matrices rebuilt, one partial update, the edge
    likelihood) and plain Lk(b) are served by the resident form of traverse_aa_kernel -- the workgroups rebuild the queued matrices
    themselves -- instead of pmat20_kernel + traverse_aa_kernel launches.  Every scalar is the launch path's double (PHYHIP_RESIDENT=0),
    the oracle's to 1e-11, the matrices and partial vectors left behind are the launched ones bit for bit, and the resident
    workgroups were the path that ran.  (Host-computed matrices -- the bit-exact route -- are uploads: those candidates are
    launched, the evaluations without a matrix change are still served.)"""
    from phyml_amd import replay
    from replay_oracle import OracleReplayer
    vals = {}
    for res in ("0", "1"):
        monkeypatch.setenv("PHYHIP_RESIDENT", res)
        t, ot, tree, st = synthetic_pair(16, P, 20, C, seed=41, ambiguous_every=9, host_pmat=host_pmat)
        try:
            t.Set_Both_Sides(True)
            ref0 = t.Lk(None)
            ot.lk(None, both_sides=True)
            tr = replay.make_trace(16, tree.edge_left, tree.edge_rght, tree.edge_len, 80, seed=13, walk_every=3, opt_every=5, n_dlk=3)
            got, got2 = t.Replay_Surface_Trace(tr)
            k = tr["kind"]
            served, launches, silent, busy = t.inst.resident_stats(1)
            if res == "1" and not host_pmat:
                assert served >= 60 and silent == 0, (served, launches, silent, busy)
            # the branch-length chains' dLk calls (src/optimiz.c:607-663): resident too -- a four-word command (the edge length; the
            # workgroups build the exponential table, as the launched 20-state dlk_kernel does: the same doubles)
            d_served, _, d_silent, _ = t.inst.resident_stats(0)
            if res == "1":
                assert d_served >= 30 and d_silent == 0, (d_served, d_silent)
            if res == "0":
                assert served == 0 and d_served == 0
            # what is left in device memory: matrices (both tables feed later launches) and the buffers the stream wrote
            mats = [t.inst.get_transition_matrix(e).copy() for e in range(ot.ne)]
            bufs = [t.partials(e, s).copy() for e in (0, 5, 11, ot.ne - 1) for s in (0, 1) if (e, s) in ot.plk]
            after = t.Lk(None)   # a launched list-form traversal on whatever the residents left in the A-operand table
            if res == "1":
                oref, oref2 = OracleReplayer(ot).run(tr)
                lnl_calls = (k == replay.EDGE_LNL) | (k == replay.DLK)
                assert np.max(np.abs(got[lnl_calls] - oref[lnl_calls]) / np.abs(oref[lnl_calls])) < 1e-11
            vals[res] = (got, got2, mats, bufs, after, ref0)
        finally:
            t.close()
    a, b = vals["0"], vals["1"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert all(np.array_equal(x, y) for x, y in zip(a[2], b[2])) and all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
    assert a[4] == b[4] and a[5] == b[5]
