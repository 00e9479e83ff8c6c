#!/usr/bin/env python3
"""cfg5-style measurement (BASELINE configs[4]): incremental SPR / branch-length call pattern on a large tree.
Replays a seeded surface-call stream (phyml_amd/replay.py) through the C host layer and reports microseconds per
regraft candidate (= 3 matrix refreshes + 1 partial update + 1 edge lnL with the scalar back on the host),
candidates/s and incremental site-updates/s.  Developer / evidence tool; not the driver's bench."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--taxa", type=int, default=500)
    ap.add_argument("--patterns", type=int, default=100000)
    ap.add_argument("--candidates", type=int, default=2000)
    ap.add_argument("--opt-every", type=int, default=0)
    ap.add_argument("--states", type=int, default=4, choices=(4, 20))
    ap.add_argument("--host-pmat", action="store_true", help="transition matrices computed by the host layer's PMat() (bit-exact route)")
    ap.add_argument("--devices", default="", help="comma-separated device list: a sharded instance (repeat a device for several shards on it)")
    args = ap.parse_args()
    from phyml_amd import lktree, replay, synth, workloads
    blk = workloads.model_block("model_gtr_g4" if args.states == 4 else "model_lg_g4")
    tree = synth.random_tree(args.taxa, 9, 0.02, 0.15)   # defaults = workloads cfg5_nt_500x100k (tests/test_gpu_cfg5.py checks
    st = synth.simulate_states(tree, args.patterns, args.states, 9)  # every scalar of this call pattern against the oracle at this size)
    C = int(blk["ncatg"][0])
    t = lktree.LkTree(args.taxa, tree.edge_left, tree.edge_rght, tree.edge_len, args.patterns, args.states, C, host_pmat=args.host_pmat,
                      devices=[int(x) for x in args.devices.split(",")] if args.devices else None)
    t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"])
    t.Make_Tree_For_Lk(np.ones(args.patterns))
    t.set_tips(tip_states=st.astype(np.int32))
    t.Set_Both_Sides(True)
    t0 = time.perf_counter(); lnl = t.Lk(None); t_full = time.perf_counter() - t0
    t0 = time.perf_counter(); lnl = t.Lk(None); t_full = time.perf_counter() - t0
    tr = replay.make_trace(args.taxa, tree.edge_left, tree.edge_rght, tree.edge_len, args.candidates, seed=3, walk_every=3,
                           opt_every=args.opt_every, n_dlk=10)
    t.Replay_Surface_Trace({k: v[:200] for k, v in tr.items()})  # warm
    t0 = time.perf_counter()
    out, out2 = t.Replay_Surface_Trace(tr)
    dt = time.perf_counter() - t0
    k = tr["kind"]
    n_upd = int((k == replay.UPDATE).sum()); n_lnl = int((k == replay.EDGE_LNL).sum()); n_dlk = int((k == replay.DLK).sum())
    print(json.dumps({"matrices": "host PMat()" if args.host_pmat else "device", "states": args.states, "taxa": args.taxa, "patterns": args.patterns, "candidates": args.candidates,
                      "full_both_sides_Lk_ms": t_full * 1e3, "lnL": lnl,
                      "us_per_candidate": dt / args.candidates * 1e6, "candidates_per_s": args.candidates / dt,
                      "surface_calls": int(len(k)), "updates": n_upd, "edge_lnl": n_lnl, "dlk": n_dlk,
                      "incremental_M_site_updates_per_s": n_upd * args.patterns / dt / 1e6,
                      "finite": bool(np.isfinite(out).all()),
                      "devices": args.devices or None, "served_by_resident_workgroups": [int(t.inst.resident_stats(k)[0]) for k in (0, 1, 2)]}))
    t.close()


if __name__ == "__main__":
    main()
