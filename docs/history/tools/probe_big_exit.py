#!/usr/bin/env python3
"""tools/probe_big_exit.py: what leaving costs -- a launched evaluation (Lk(b) under profiling, i.e. never resident) right
after a phase served by the large-grid resident workgroups, against the same call with nobody resident.  One JSON line per
configuration (each in its own process).  Developer / evidence tool."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(label):
    import numpy as np
    from phyml_amd import lktree, synth, workloads
    blk = workloads.model_block("model_gtr_g4")
    taxa, P = 100, 100000
    tree = synth.random_tree(taxa, 9, 0.02, 0.15)
    st = synth.simulate_states(tree, P, 4, 9)
    t = lktree.LkTree(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, P, 4, 4)
    t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"])
    t.Make_Tree_For_Lk(np.ones(P)); t.set_tips(tip_states=st.astype(np.int32)); t.Set_Both_Sides(True); t.Lk(None)
    e = t.ne // 2
    t.Lk(e); t.Update_Eigen_Lr(e); t.Set_Update_Eigen_Lr(0); t.Set_Use_Eigen_Lr(1)
    after_res, alone = [], []
    for rep in range(30):
        for i in range(6):
            t.dLk(0.05 + 1e-3 * i, e)          # resident phase
        t0 = time.perf_counter(); t.inst.synchronize(); after_res.append((time.perf_counter() - t0) * 1e6)  # release + wait for their exit
        t0 = time.perf_counter(); t.inst.synchronize(); alone.append((time.perf_counter() - t0) * 1e6)
    s = t.inst.resident_stats(2)
    t.close()
    print(json.dumps({"label": label, "us_synchronize_after_resident_phase_median": float(np.median(after_res)),
                      "max": float(np.max(after_res)), "us_synchronize_alone_median": float(np.median(alone)), "big_resident": list(s)}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--label":
        worker(sys.argv[2]); sys.exit(0)
    diag = os.path.join(ROOT, "phyml_amd", "lib_diag")
    for name, env in (("product", {}), ("tickets", {"PHYHIP_LIBDIR": diag, "PHYHIP_BIG_GROUP_SUM": "0"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--label", name], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else json.dumps({"label": name, "error": r.stderr[-600:]}), flush=True)
