"""Wall time per Update_Eigen_Lr and per dLk (K3 / K4) at a given size: python tools/bench_dlk.py [patterns] [aa]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from phyml_amd import lktree, workloads
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
aa = len(sys.argv) > 2 and sys.argv[2] == "aa"
wl = workloads.make("cfg3_aa_200x10k" if aa else "cfg2_nt_100x50k", n_pattern=P)
tree, st, blk = wl["tree"], wl["states"], wl["model"]
n, S = st.shape[0], (20 if aa else 4)
t = lktree.LkTree(n, tree.edge_left, tree.edge_rght, tree.edge_len, P, S, 4, host_pmat=False)
t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"],
            float(blk["l_min"][0]), float(blk["l_max"][0]))
t.Make_Tree_For_Lk(np.ones(P)); t.set_tips(tip_states=st.astype(np.int32))
t.Set_Both_Sides(1); t.Lk()
e = t.ne // 2
t.Lk(e)
reps = 200
t0 = time.perf_counter()
for _ in range(reps): t.Update_Eigen_Lr(e); t.inst.L.phyhip_synchronize(t.inst.id)
t_eig = (time.perf_counter() - t0) / reps
t.Set_Update_Eigen_Lr(0); t.Set_Use_Eigen_Lr(1)
for i in range(5): t.dLk(0.05, e)  # (first launches load code)
t0 = time.perf_counter()
for i in range(reps): t.dLk(0.05 + 1e-4 * i, e)
t_dlk = (time.perf_counter() - t0) / reps
t.Set_Use_Eigen_Lr(0)
t_lk = 0.0
if not os.environ.get("BENCH_DLK_ONLY"):  # (PHYHIP_RESIDENT_STATS: the large-grid evaluator's means then describe the dLk commands alone)
    t0 = time.perf_counter()
    for i in range(reps): t.Lk(e)
    t_lk = (time.perf_counter() - t0) / reps
b = P * 4 * S * 8
print(json.dumps({"patterns": P, "states": S, "us_Update_Eigen_Lr": t_eig * 1e6, "us_dLk": t_dlk * 1e6, "us_Lk_edge": t_lk * 1e6,
                  "dot_prod_MB": b / 1e6, "dLk_stream_GBps": b / t_dlk / 1e9}))
t.close()
