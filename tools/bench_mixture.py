"""Throughput of a 4-class mixture evaluation (LG4X-shaped: four class trees, one category each) on the device:
full post-order of every class tree + phyhip_calculate_mixture_log_likelihood, synthetic 200 taxa x P aa patterns."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from phyml_amd import capi, lktree, workloads

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
K = 4
wl = workloads.make("cfg3_aa_200x10k", n_pattern=P)
tree, st, blk = wl["tree"], wl["states"], wl["model"]
n = tree.n_otu if hasattr(tree, "n_otu") else len(st)
rates = [0.2, 0.75, 1.95, 5.16]; proba = [0.42, 0.34, 0.18, 0.06]
trees = []
for k in range(K):
    t = lktree.LkTree(n, tree.edge_left, tree.edge_rght, tree.edge_len, P, 20, 1, host_pmat=False)
    t.set_model(blk["pi"], np.array([rates[k]]), np.array([1.0]), blk["e_val"], blk["r_e_vect"], blk["l_e_vect"],
                float(blk["l_min"][0]), float(blk["l_max"][0]), 1.0, 1, 0, 0.0)
    t.Make_Tree_For_Lk(np.ones(P), None)
    t.set_tips(tip_states=[st[i].astype(np.int32) for i in range(n)])
    trees.append(t)
dummy = torch.zeros(8, dtype=torch.float64, device="cuda")
e = None
def step():
    for t in trees:
        t.Lk_Shard_Device(dummy.data_ptr())
    ids = [t.tree.contents.b_inst for t in trees]
    b = trees[0].root_edge() if hasattr(trees[0], "root_edge") else 0
    return capi.mixture_log_likelihood(ids, [t.side_buffer(re, 0) for t in trees], [t.side_buffer(re, 1) for t in trees],
                                       [t.edge(re).contents.Pij_rr_idx for t in trees], proba, [1.0] * K, [1.0] * K, float(K), float(K), 1.0 / K)
# root edge of the host layer = edge of tip_root
t0 = trees[0]
re = t0.node(t0.tip_root).contents.b[0].contents.num
lnl = step()
for _ in range(5): step()
torch.cuda.synchronize()
reps = 30
t_start = time.perf_counter()
for _ in range(reps): lnl = step()
dt = (time.perf_counter() - t_start) / reps
upd = float(P) * (n - 2) * K
print(json.dumps({"patterns": P, "taxa": n, "classes": K, "lnL": lnl, "ms_per_mixture_eval": dt * 1e3,
                  "class_site_updates_per_s_M": upd / dt / 1e6,
                  "equivalent_4cat_site_updates_per_s_M": float(P) * (n - 2) / dt / 1e6}))
for t in trees: t.close()
