#!/usr/bin/env python3
"""tools/pmc_cfg5.py <kind> [--patterns P] [--taxa N] [--n K]: ONE call kind of the cfg5 call pattern (500 taxa x 100 000 nt patterns) as
kernel LAUNCHES, for counter profiles (tools/profile_r05_extra.sh runs it under `rocprofv3 --pmc ...`): with PHYHIP_RESIDENT=0 every
scalar-returning call is a dispatch -- the large-grid kernel's one-shot form (phyhip_big.hpp, BigArgs::n_one_shot: the same tile
bodies the resident workgroups run), `eigen_lr_kernel`, `pmat_kernel`.
  spr   K regraft candidates (3 matrix refreshes + 1 partial update + the edge lnL)
  dlk   K dLk calls behind one Update_Eigen_Lr
  eig   K Update_Eigen_Lr calls
  full  K both-sides Lk(NULL)"""
import argparse
import os
import sys

os.environ.setdefault("PHYHIP_RESIDENT", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from phyml_amd import lktree, replay, synth, workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("kind", choices=["spr", "dlk", "eig", "full"])
ap.add_argument("--patterns", type=int, default=100000)
ap.add_argument("--taxa", type=int, default=500)
ap.add_argument("--n", type=int, default=40)
a = ap.parse_args()
blk = workloads.model_block("model_gtr_g4")
tree = synth.random_tree(a.taxa, 9, 0.02, 0.15)
st = synth.simulate_states(tree, a.patterns, 4, 9)
t = lktree.LkTree(a.taxa, tree.edge_left, tree.edge_rght, tree.edge_len, a.patterns, 4, int(blk["ncatg"][0]))
t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"])
t.Make_Tree_For_Lk(np.ones(a.patterns))
t.set_tips(tip_states=st.astype(np.int32))
t.Set_Both_Sides(True)
t.Lk(None)
if a.kind == "full":
    for _ in range(a.n):
        t.Lk(None)
elif a.kind == "spr":
    tr = replay.make_trace(a.taxa, tree.edge_left, tree.edge_rght, tree.edge_len, a.n, seed=3, walk_every=0, opt_every=0, n_dlk=0)
    t.Replay_Surface_Trace(tr)
else:
    e = t.ne // 2
    t.Lk(e)
    t.Update_Eigen_Lr(e)
    t.Set_Update_Eigen_Lr(0); t.Set_Use_Eigen_Lr(1)
    if a.kind == "dlk":
        for i in range(a.n):
            t.dLk(0.05 + 1e-4 * i, e)
    else:
        for i in range(a.n):
            t.Set_Update_Eigen_Lr(1); t.Update_Eigen_Lr(e); t.Set_Update_Eigen_Lr(0)
t.close()
print("PMC_CFG5_DONE", a.kind)
