"""Helpers shared by the GPU parity tests: build a device LkTree (C host layer) from a golden dump."""
import numpy as np

import orc
from phyml_amd import lktree


def device_tree_from_golden(d, host_pmat=True, devices=None, force_sharded=False, use_m4mod=False, arith=1):
    """Device tree with the reference's own neighbour order (node_v/node_b of the dump); tips come from the
    oracle's tip encoder.  host_pmat=True: the C host layer's own PMat() + upload (bit-exact route, src/lk.c:2360);
    False: device PMat from the eigen system (src/lk.c:2344)."""
    ot = orc.tree_from_golden(d, arith=arith)
    m = ot.m
    t = lktree.LkTree(ot.n, d["edge_left"], d["edge_rght"], d["edge_len"], ot.P, m.ns, m.ncatg,
                      node_v=d["node_v"], node_b=d["node_b"], host_pmat=host_pmat, devices=devices, force_sharded=force_sharded,
                      use_m4mod=use_m4mod)
    t.tip_root = ot.tip_root
    t.set_model(m.pi, m.gamma_rr, m.gamma_r_proba, m.e_val, m.r_e_vect, m.l_e_vect, m.l_min, m.l_max, m.br_len_mult,
                int(d["apply_lk_scaling"][0]), m.invar_model, m.pinvar)
    t.Make_Tree_For_Lk(d["wght"], d["invar"])
    t.set_tips(tip_partials=ot.tip_vec)
    return t, ot


def synthetic_pair(n_otu, P, ns, C, seed, lmin=0.02, lmax=0.3, wght=None, apply_scaling=1, host_pmat=True, ambiguous_every=0,
                   devices=None, force_sharded=False, use_m4mod=False, arith=1):
    """Device tree + oracle tree on a seeded synthetic alignment with C rate classes (rates/weights made up,
    normalised) and the committed model block's eigen system."""
    from phyml_amd import synth, workloads
    blk = dict(workloads.model_block("model_gtr_g4" if ns == 4 else "model_lg_g4"))
    rates = np.linspace(0.2, 2.2, C) if C > 1 else np.array([1.0])
    w = np.linspace(1.0, 2.0, C); w = w / w.sum()
    rates = rates / float((rates * w).sum())
    blk["ncatg"] = np.array([float(C)]); blk["gamma_rr"] = rates; blk["gamma_r_proba"] = w
    m = orc.Model(blk)
    tree = synth.random_tree(n_otu, seed, lmin, lmax)
    st = synth.simulate_states(tree, P, ns, seed)
    chars = synth.states_to_chars(st, ns).copy()
    if ambiguous_every:
        amb = (b"N-RY?" if ns == 4 else b"X-?BZ")
        for t in range(n_otu):
            idx = np.arange((t * 7) % ambiguous_every, P, ambiguous_every)
            chars[t, idx] = np.frombuffer(amb, dtype=np.uint8)[(idx + t) % len(amb)]
    wg = np.ones(P) if wght is None else np.asarray(wght, dtype=np.float64)
    tv, ds, amb_ = [], [], []
    for t in range(n_otu):
        v, s, a = orc.init_tip(m.datatype, chars[t])
        tv.append(v); ds.append(s); amb_.append(a)
    ot = orc.OracleTree(m, n_otu, tree.edge_left, tree.edge_rght, tree.edge_len, wg, tv, ds, amb_, apply_scaling=apply_scaling, arith=arith)
    t = lktree.LkTree(n_otu, tree.edge_left, tree.edge_rght, tree.edge_len, P, ns, C, host_pmat=host_pmat, devices=devices,
                      force_sharded=force_sharded, use_m4mod=use_m4mod)
    t.set_model(m.pi, m.gamma_rr, m.gamma_r_proba, m.e_val, m.r_e_vect, m.l_e_vect, m.l_min, m.l_max, 1.0, apply_scaling)
    t.Make_Tree_For_Lk(wg)
    t.set_tips(tip_partials=tv)
    return t, ot, tree, st
