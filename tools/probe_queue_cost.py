#!/usr/bin/env python3
"""tools/probe_queue_cost.py: the host's share of an SPR candidate at the cfg5 size -- the candidate stream of tools/bench_big.py
with its evaluations taken out (three matrix refreshes + the partial update(s) only queue), against the full stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from phyml_amd import lktree, replay, synth, workloads

blk = workloads.model_block("model_gtr_g4")
taxa, P = 500, 100000
tree = synth.random_tree(taxa, 9, 0.02, 0.15)
st = synth.simulate_states(tree, P, 4, 9)
t = lktree.LkTree(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, P, 4, 4)
t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"])
t.Make_Tree_For_Lk(np.ones(P)); t.set_tips(tip_states=st.astype(np.int32)); t.Set_Both_Sides(True); t.Lk(None)
tr = replay.make_trace(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, 600, seed=3, walk_every=3, opt_every=0, n_dlk=5)
t.Replay_Surface_Trace({k: v[:300] for k, v in tr.items()})
t0 = time.perf_counter(); t.Replay_Surface_Trace(tr); full = (time.perf_counter() - t0) / 600 * 1e6
keep = tr["kind"] != replay.EDGE_LNL
q = {k: v[keep][:5 * 150] for k, v in tr.items()}   # 150 candidates' worth of queueing calls (no evaluation: nothing launches)
best = 1e9
for rep in range(5):
    t0 = time.perf_counter(); t.Replay_Surface_Trace(q); dt = (time.perf_counter() - t0) / 150 * 1e6
    t.Lk(None)  # empties the queue
    best = min(best, dt)
print({"us_per_candidate": round(full, 2), "us_of_it_queueing_calls_only": round(best, 2)})
t.close()
