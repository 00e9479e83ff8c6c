"""BASELINE configs[4] at full size (cfg5: 500 taxa x 100 000 nt patterns, GTR+G4): the incremental call pattern of
spr.c / optimiz.c -- regraft candidates (three matrix refreshes, Update_Partial_Lk into a spare buffer, Lk(b)), path
updates, Update_Eigen_Lr + dLk series (src/spr.c:543,643-646; src/optimiz.c:622-632) -- replayed on the device, EVERY
returned lnL and dlnL checked against the CPU restatement on the same stream at the same size.

The oracle cannot hold 1494 partial vectors of 100 000 patterns at once (19 GB), but lnL and dlnL are sums over patterns
(src/lk.c:744-745,856): it replays the whole stream on pattern chunks in worker processes and the per-call sums over the
chunks are what the device must have returned (1e-10 relative; the device adds 100 000 terms in another order)."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = "cfg5_nt_500x100k"
CHUNK = 2500


def _oracle_chunk(args):
    lo, n, trace = args
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from phyml_amd import synth, workloads
    from replay_oracle import OracleReplayer
    wl = workloads.make(CFG, n_pattern=n, pattern_offset=lo)
    tree, st = wl["tree"], wl["states"]
    m = orc.Model(wl["model"])
    chars = synth.states_to_chars(st, 4)
    tv, ds, amb = [], [], []
    for t in range(tree.n_otu):
        v, s, a = orc.init_tip(m.datatype, chars[t])
        tv.append(v); ds.append(s); amb.append(a)
    ot = orc.OracleTree(m, tree.n_otu, tree.edge_left, tree.edge_rght, tree.edge_len, np.ones(n), tv, ds, amb)
    full = ot.lk(None, both_sides=True)
    out, out2 = OracleReplayer(ot).run(trace)
    return full, out, out2


def test_cfg5_call_stream_parity_at_full_size():
    from phyml_amd import lktree, replay, workloads
    cfg = workloads.CONFIGS[CFG]
    P, n = cfg["n_pattern"], cfg["n_otu"]
    wl = workloads.make(CFG)
    tree, st, blk = wl["tree"], wl["states"], wl["model"]
    tr = replay.make_trace(n, tree.edge_left, tree.edge_rght, tree.edge_len, 160, seed=3, walk_every=3, opt_every=5, n_dlk=4)
    k = tr["kind"]
    lnl_calls = (k == replay.EDGE_LNL) | (k == replay.DLK)
    assert (k == replay.EDGE_LNL).sum() == 160 and (k == replay.DLK).sum() == 128

    t = lktree.LkTree(n, tree.edge_left, tree.edge_rght, tree.edge_len, P, 4, 4, host_pmat=True)
    try:
        t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"],
                    float(blk["l_min"][0]), float(blk["l_max"][0]))
        t.Make_Tree_For_Lk(np.ones(P))
        t.set_tips(tip_states=st.astype(np.int32))
        t.Set_Both_Sides(True)
        full = t.Lk(None)
        got, got2 = t.Replay_Surface_Trace(tr)
        again = t.Lk(None)   # the stream leaves the tree's own buffers intact
    finally:
        t.close()

    jobs = [(lo, min(CHUNK, P - lo), tr) for lo in range(0, P, CHUNK)]
    with mp.get_context("spawn").Pool(min(16, os.cpu_count() or 1)) as pool:
        res = pool.map(_oracle_chunk, jobs)
    ref_full = sum(r[0] for r in res)
    ref = np.sum([r[1] for r in res], axis=0)
    ref2 = np.sum([r[2] for r in res], axis=0)
    assert abs(full - ref_full) / abs(ref_full) < 1e-10
    assert again == full
    assert np.max(np.abs(got[lnl_calls] - ref[lnl_calls]) / np.abs(ref[lnl_calls])) < 1e-10
    dl = k == replay.DLK
    assert np.max(np.abs(got2[dl] - ref2[dl]) / np.maximum(1.0, np.abs(ref2[dl]))) < 1e-8


# ---- a RECORDED search at the cfg5 pattern count (round 4) ------------------------------------------------------------------
# tests/golden/trace_synth200_spr.phyg: the first 46 000 surface calls of PhyML's own SPR search on a 200-taxon alignment
# (oracle/trace_driver.c, tests/golden/make_traces.py) -- four rounds of Br_Len_Opt over every edge, then the SPR phase
# (src/spr.c:149,813).  The reference cannot run 100 000 patterns x 200 taxa in one piece either (its slab is sized with an
# `int`), so the call STREAM -- buffers, matrices, lengths: what spr.c / optimiz.c decided -- is replayed on a 100 000-pattern
# alignment of the same 200 taxa, on the device and, in 2 500-pattern chunks, on the pinned restatement (which reproduces every
# scalar of the recording itself exactly, tests/test_trace_oracle.py): the per-call sums over the chunks are what the device
# must return.  Replaces the seeded stream above as cfg5's call-pattern pin where the two overlap.
REC = "trace_synth200_spr"
REC_P = 100000
REC_WINDOWS = ((0, 6000), (36000, 46000))  # Br_Len_Opt rounds; the SPR phase (replayed from the start: the state builds up)


def _recorded_inputs(lo, n):
    from phyml_amd import phyg, synth
    d = phyg.load(os.path.join(ROOT, "tests", "golden", REC + ".phyg"))
    tree = synth.random_tree(int(d["n_otu"][0]), 41, 0.02, 0.12)
    st = synth.simulate_states(tree, n, 4, 77, site_offset=lo)  # (every column depends only on its own index)
    return d, st


def _recorded_chunk(args):
    lo, n, upto = args
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from phyml_amd import replay, synth
    import replay_oracle
    d, st = _recorded_inputs(lo, n)
    tr, _, _ = replay.recorded_trace(d)
    tr = {k: v[:upto] for k, v in tr.items()}
    m = orc.Model(d)
    chars = synth.states_to_chars(st, 4)
    tv, ds, amb = [], [], []
    for t in range(st.shape[0]):
        v, s, a = orc.init_tip(m.datatype, chars[t])
        tv.append(v); ds.append(s); amb.append(a)
    ot = orc.OracleTree(m, st.shape[0], d["edge_left"], d["edge_rght"], d["edge_len"], np.ones(n), tv, ds, amb,
                        apply_scaling=int(d["apply_lk_scaling"][0]))
    return replay_oracle.RecordedReplayer(ot).run(tr)


def test_recorded_200_taxon_search_at_100k_patterns():
    from phyml_amd import lktree, replay
    d, st = _recorded_inputs(0, REC_P)
    tr, _, _ = replay.recorded_trace(d)
    upto = max(hi for _, hi in REC_WINDOWS)
    tr = {k: v[:upto] for k, v in tr.items()}
    n = int(d["n_otu"][0])
    t = lktree.LkTree(n, d["edge_left"], d["edge_rght"], d["edge_len"], REC_P, 4, int(d["ncatg"][0]), host_pmat=False)
    try:
        t.set_model(d["pi"], d["gamma_rr"], d["gamma_r_proba"], d["e_val"], d["r_e_vect"], d["l_e_vect"], float(d["l_min"][0]),
                    float(d["l_max"][0]), float(d["br_len_mult"][0]), int(d["apply_lk_scaling"][0]))
        t.Make_Tree_For_Lk(np.ones(REC_P))
        t.set_tips(tip_states=st.astype(np.int32))
        got, got2 = t.Replay_Surface_Trace(tr)
        served = t.inst.resident_stats(2)[0]
    finally:
        t.close()
    jobs = [(lo, min(CHUNK, REC_P - lo), upto) for lo in range(0, REC_P, CHUNK)]
    with mp.get_context("spawn").Pool(min(16, os.cpu_count() or 1)) as pool:
        res = pool.map(_recorded_chunk, jobs)
    ref = np.sum([r[0] for r in res], axis=0)
    ref2 = np.sum([r[1] for r in res], axis=0)
    k = tr["kind"]
    for lo, hi in REC_WINDOWS:
        w = np.zeros(len(k), bool); w[lo:hi] = True
        sc = w & np.isin(k, (replay.EDGE_LNL, replay.DLK))
        assert sc.sum() > 1000
        assert np.max(np.abs(got[sc] - ref[sc]) / np.abs(ref[sc])) < 1e-9   # (device-built matrices: device exp)
        dl = w & (k == replay.DLK)
        if dl.any():
            assert np.max(np.abs(got2[dl] - ref2[dl]) / np.maximum(1.0, np.abs(ref2[dl]))) < 1e-6
    # most of these calls never were a kernel launch: the large-grid resident evaluator (phyhip_big.hpp) served them
    assert served > 0.8 * int(np.isin(k, (replay.EDGE_LNL, replay.DLK, replay.EIGEN_LR)).sum()), served
