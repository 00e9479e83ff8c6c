#!/usr/bin/env python3
"""tools/bench_mixture_groups.py [K] [P] [taxa]: a mixture of K amino-acid classes (profile mixtures of the C10-C60 kind: one class tree
per profile, src/mixt.c:730-1160) evaluated two ways -- one instance per class (K traversal launches per MIXT_Lk) and in groups of four
classes on the category axes of ceil(K / 4) instances (include/phyhip.h: phyhip_calculate_mixture_log_likelihood, "groups of classes").
Full post-order of every class + the combination per evaluation; prints one JSON line.  The two forms are held to each other's lnL."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from phyml_amd import capi, lktree, synth, workloads

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
blk = workloads.model_block("model_lg_g4")
tree = synth.random_tree(n, 4, 0.02, 0.15)
st = synth.simulate_states(tree, P, 20, 4)
rates = 0.3 + 0.17 * np.arange(K)
proba = np.linspace(1.0, 2.0, K); proba /= proba.sum()


def make(group_rates):
    g = len(group_rates)
    t = lktree.LkTree(n, tree.edge_left, tree.edge_rght, tree.edge_len, P, 20, g, host_pmat=False, class_axis=g > 1 or None)
    t.set_model(blk["pi"], np.asarray(group_rates), np.full(g, 1.0 / g), blk["e_val"], blk["r_e_vect"], blk["l_e_vect"],
                float(blk["l_min"][0]), float(blk["l_max"][0]), 1.0, 1, 0, 0.0)
    t.Make_Tree_For_Lk(np.ones(P), None)
    if g > 1:
        for k in range(g):
            t.inst.set_state_frequencies(blk["pi"], index=k)
            t.inst.set_eigen_decomposition(blk["r_e_vect"], blk["l_e_vect"], blk["e_val"], index=k)
    t.set_tips(tip_states=[st[i].astype(np.int32) for i in range(n)])
    for e in range(t.ne):
        t.Update_PMat_At_Given_Edge(e)
    return t


def run(groups):
    trees = [make(g) for g in groups]
    root = trees[0].node(0).contents.v[0].contents.num
    re = trees[0].node(0).contents.b[0].contents.num
    ids = [t.tree.contents.b_inst for t in trees]
    par = [t.side_buffer(re, 0) for t in trees]; chi = [t.side_buffer(re, 1) for t in trees]
    pms = [t.edge(re).contents.Pij_rr_idx for t in trees]

    def step():
        for t in trees:
            t.Post_Order_Lk(0, root)
        return capi.mixture_log_likelihood_classes(ids, par, chi, pms, proba, [1.0] * K, [1.0] * K, float(K), float(K), 1.0)
    lnl = step()
    for _ in range(3):
        step()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        lnl = step()
    dt = (time.perf_counter() - t0) / reps
    for t in trees:
        t.close()
    return lnl, dt


one = run([[r] for r in rates])
grp = run([list(rates[i:i + 4]) for i in range(0, K, 4)])
print(json.dumps({"classes": K, "patterns": P, "taxa": n, "states": 20,
                  "one_instance_per_class": {"launches_per_evaluation": K, "ms_per_mixture_eval": one[1] * 1e3, "lnL": one[0]},
                  "groups_of_four_on_the_class_axis": {"launches_per_evaluation": (K + 3) // 4, "ms_per_mixture_eval": grp[1] * 1e3, "lnL": grp[0]},
                  "speedup": one[1] / grp[1], "lnL_rel_diff": abs(one[0] - grp[0]) / abs(one[0])}))
