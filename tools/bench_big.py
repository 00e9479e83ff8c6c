#!/usr/bin/env python3
"""tools/bench_big.py [--patterns P] [--taxa N] [--candidates K] : the launch-bound call patterns at the cfg5 size -- SPR regraft
candidates (3 matrix refreshes + 1 partial update + the edge lnL, scalar on the host), Br_Len_Opt chains (1 Update_Eigen_Lr +
5 dLk) and single dLk calls -- under several configurations of the engine, each in its own process (the switches are read at
instance creation): kernel launch per call (PHYHIP_RESIDENT=0), the large-grid resident evaluator with the tile sums added by
the host, and with the final sum on the device.  Prints one JSON line per configuration.  Developer / evidence tool."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    import numpy as np
    from phyml_amd import lktree, replay, synth, workloads
    blk = workloads.model_block("model_gtr_g4")
    taxa, P = args.taxa, args.patterns
    tree = synth.random_tree(taxa, 9, 0.02, 0.15)
    st = synth.simulate_states(tree, P, 4, 9)
    t = lktree.LkTree(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, P, 4, int(blk["ncatg"][0]), host_pmat=args.host_pmat)
    t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"])
    t.Make_Tree_For_Lk(np.ones(P))
    t.set_tips(tip_states=st.astype(np.int32))
    t.Set_Both_Sides(True)
    t.Lk(None)
    tr = replay.make_trace(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, args.candidates, seed=3, walk_every=3, opt_every=0, n_dlk=5)
    t.Replay_Surface_Trace({k: v[:300] for k, v in tr.items()})
    t0 = time.perf_counter()
    res, _ = t.Replay_Surface_Trace(tr)
    spr_us = (time.perf_counter() - t0) / args.candidates * 1e6
    if args.end == "spr":  # (the candidate phase alone: PHYHIP_HOSTPROF / PHYHIP_RESIDENT_STATS then describe candidates only)
        t.close()
        print(json.dumps({"label": args.label, "us_per_spr_candidate": spr_us}))
        return
    # Br_Len_Opt chains
    e = t.ne // 2
    t.Lk(e)
    t.Update_Eigen_Lr(e)
    t.Set_Update_Eigen_Lr(0); t.Set_Use_Eigen_Lr(1)
    for i in range(6):
        t.dLk(0.05, e)
    n_chain, n_dlk = 60, 5
    t0 = time.perf_counter()
    for k in range(n_chain):
        t.Set_Update_Eigen_Lr(1); t.Update_Eigen_Lr(e); t.Set_Update_Eigen_Lr(0)
        for i in range(n_dlk):
            t.dLk(0.05 + 1e-3 * i + 1e-5 * k, e)
    chain_us = (time.perf_counter() - t0) / n_chain * 1e6
    t0 = time.perf_counter()
    vals = [t.dLk(0.05 + 1e-4 * i, e)[1] for i in range(300)]
    dlk_us = (time.perf_counter() - t0) / 300 * 1e6
    t0 = time.perf_counter()
    for k in range(100):
        t.Set_Update_Eigen_Lr(1); t.Update_Eigen_Lr(e); t.Set_Update_Eigen_Lr(0)
    eig_us = (time.perf_counter() - t0) / 100 * 1e6
    if args.end == "dlk":
        for i in range(50):
            t.dLk(0.05 + 1e-4 * i, e)
    t.Set_Use_Eigen_Lr(0)
    if args.end == "spr":
        t.Replay_Surface_Trace({k: v[:304] for k, v in tr.items()})
    out = {"label": args.label, "patterns": P, "taxa": taxa, "us_per_spr_candidate": spr_us, "us_per_chain_1_eigen_5_dlk": chain_us,
           "us_per_dlk": dlk_us, "us_per_update_eigen_lr": eig_us, "checksum_spr": float(np.sum(res)), "checksum_dlk": float(np.sum(vals)),
           "big_resident": t.inst.resident_stats(2), "small_resident": (t.inst.resident_stats(0), t.inst.resident_stats(1))}
    t.close()
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--taxa", type=int, default=500)
    ap.add_argument("--patterns", type=int, default=100000)
    ap.add_argument("--candidates", type=int, default=600)
    ap.add_argument("--host-pmat", action="store_true")
    ap.add_argument("--label", default=None)
    ap.add_argument("--end", default="eig", help="kind of the last command (PHYHIP_RESIDENT_STATS prints the last command's stamps)")
    ap.add_argument("--configs", default="launch,host_sum,device_sum,product")
    args = ap.parse_args()
    if args.label is not None:
        return worker(args)
    diag = os.path.join(ROOT, "phyml_amd", "lib_diag")
    cfgs = {"launch": {"PHYHIP_RESIDENT": "0"}, "product": {},
            "host_sum": {"PHYHIP_LIBDIR": diag, "PHYHIP_BIG_DEVICE_SUM": "100000000"},
            "device_sum": {"PHYHIP_LIBDIR": diag, "PHYHIP_BIG_DEVICE_SUM": "0"},
            "tickets": {"PHYHIP_LIBDIR": diag, "PHYHIP_BIG_GROUP_SUM": "0"},  # device sum through per-tile sums and tickets
            "diag": {"PHYHIP_LIBDIR": diag},
            "launch_old": {"PHYHIP_LIBDIR": diag, "PHYHIP_RESIDENT": "0", "PHYHIP_BIG_ONE_SHOT": "0"},  # round 3's launch per call
            "g1": {"PHYHIP_LIBDIR": diag, "PHYHIP_NT_GROUPS": "1"}, "g2": {"PHYHIP_LIBDIR": diag, "PHYHIP_NT_GROUPS": "2"},
            "g1_launch": {"PHYHIP_LIBDIR": diag, "PHYHIP_NT_GROUPS": "1", "PHYHIP_RESIDENT": "0"}}
    for name in args.configs.split(","):
        if name.startswith("push"):  # pushN: command records in device memory (1 hipMalloc, 2 fine-grained, 3 uncached), diag build
            cfgs[name] = {"PHYHIP_LIBDIR": diag, "PHYHIP_PUSH_CMDS": name[4:]}
        if name.startswith("lib_"):  # another build of the engine (tools/build_variant.sh <name>): phyml_amd/lib_<name>
            cfgs[name] = {"PHYHIP_LIBDIR": os.path.join(ROOT, "phyml_amd", name)}
        env = dict(os.environ); env.update(cfgs[name])
        cmd = [sys.executable, os.path.abspath(__file__), "--label", name, "--taxa", str(args.taxa), "--patterns", str(args.patterns),
               "--candidates", str(args.candidates)] + (["--host-pmat"] if args.host_pmat else [])
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else json.dumps({"label": name, "error": r.stderr[-400:]})
        print(line, flush=True)


if __name__ == "__main__":
    main()
