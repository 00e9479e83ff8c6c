"""cfg5 SPR candidate latency, several timed replays in ONE process (and in several processes): is the 30 / 36 us split a
property of a process (placement of its buffers) or of time (clocks)?"""
import os, sys, time, json
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
out = []
for rep in range(6):
    tr = replay.make_trace(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, 500, seed=3 + rep, walk_every=3, opt_every=0, n_dlk=5)
    t0 = time.perf_counter(); t.Replay_Surface_Trace(tr); dt = time.perf_counter() - t0
    out.append(round(dt / 500 * 1e6, 1))
print(out)
t.close()
