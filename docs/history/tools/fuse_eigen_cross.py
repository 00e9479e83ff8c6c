#!/usr/bin/env python3
"""Update_Eigen_Lr as the last step of a traversal launch (lane-per-pattern nucleotide kernel) against eigen_lr_kernel, by
pattern count: wall time of Br_Len_Opt's chain -- [queued partial updates at the edge] Update_Eigen_Lr, 5 dLk -- with and
without partial updates queued in front.  No profiling (event records would be timed too).  Run once per build
(PHYHIP_LIBDIR): the product limits the fused form to PHYHIP_FUSE_EIGEN_MAX patterns."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phyml_amd import lktree, synth, workloads  # noqa: E402
import numpy as np  # noqa: E402

blk = workloads.model_block("model_gtr_g4")
C = int(blk["ncatg"][0])
for P in (382, 2048, 4096, 8192, 16384, 32768, 65536, 131072):
    tree = synth.random_tree(40, 9, 0.02, 0.15)
    st = synth.simulate_states(tree, P, 4, 9)
    t = lktree.LkTree(40, tree.edge_left, tree.edge_rght, tree.edge_len, P, 4, C, host_pmat=False)
    t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"])
    t.Make_Tree_For_Lk(np.ones(P))
    t.set_tips(tip_states=st.astype(np.int32))
    t.Set_Both_Sides(True)
    t.Lk(None)
    out = {"patterns": P}
    for queued in (0, 1):
        edges = [5, 17, 23, 31, 44, 52, 60, 9]
        def chain(e, k):
            if queued:
                t.Update_Lk_At_Given_Edge(e)   # the partial updates Br_Len_Opt's traversal leaves queued at the edge
            t.Set_Update_Eigen_Lr(1); t.Update_Eigen_Lr(e); t.Set_Update_Eigen_Lr(0)
            t.Set_Use_Eigen_Lr(1)
            for i in range(5):
                t.dLk(0.05 + 1e-3 * i + 1e-5 * k, e)
            t.Set_Use_Eigen_Lr(0)
        for k in range(16):
            chain(edges[k % 8], k)
        n = 200
        t0 = time.perf_counter()
        for k in range(n):
            chain(edges[k % 8], k)
        out["chain_us_queued%d" % queued] = round((time.perf_counter() - t0) / n * 1e6, 1)
    print(json.dumps(out))
    t.close()
