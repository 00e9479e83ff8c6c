#!/usr/bin/env python3
"""What the collective path costs one evaluation: Lk(b) on a small alignment (the traversal is a few
microseconds) on a plain instance against sharded instances with 1, 2 and 8 shards -- all on device 0 when the box has one
GPU (per-shard launches + fixed-order local sum + one-rank ncclAllReduce + publish), on distinct devices when it has more.
Prints one JSON line: microseconds per evaluation for each layout and the difference to the plain instance."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch
    from phyml_amd import workloads
    import bench
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    reps = 2000
    wl = workloads.make("cfg2_nt_100x50k", n_pattern=P)
    ndev = torch.cuda.device_count()
    rows = {}
    layouts = [("plain", None), ("1_shard", [0]), ("2_shards", [0, 0] if ndev < 2 else [0, 1]),
               ("8_shards", [0] * 8 if ndev < 8 else list(range(8)))]
    for name, devs in layouts:
        for threads in ((None,) if devs is None else ("0", "1")):
            if threads is not None:
                os.environ["PHYHIP_SHARD_THREADS"] = threads
            if devs is None:
                t = bench.build_tree(wl, device=0)
            else:
                from phyml_amd import lktree
                tree, st, blk = wl["tree"], wl["states"], wl["model"]
                t = lktree.LkTree(tree.n_otu, tree.edge_left, tree.edge_rght, tree.edge_len, P, 4, int(blk["ncatg"][0]), devices=devs,
                                  force_sharded=True)
                t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"],
                            float(blk["l_min"][0]), float(blk["l_max"][0]), 1.0, 1)
                t.Make_Tree_For_Lk(np.ones(P))
                t.set_tips(tip_states=wl["states"].astype(np.int32))
            t.Set_Both_Sides(True)
            ref = t.Lk(None)
            e = 3
            for _ in range(200):
                v = t.Lk(e)
            t0 = time.perf_counter()
            for _ in range(reps):
                v = t.Lk(e)
            dt = (time.perf_counter() - t0) / reps
            assert abs(v - ref) / abs(ref) < 1e-10, (v, ref)
            rows[name + ("" if threads is None else f"_threads{threads}")] = {"us_per_Lk_b": dt * 1e6, "devices": devs}
            t.close()
    base = rows["plain"]["us_per_Lk_b"]
    for k, r in rows.items():
        r["over_plain_us"] = r["us_per_Lk_b"] - base
    print(json.dumps({"patterns": P, "taxa": wl["tree"].n_otu, "devices_visible": ndev, "rows": rows}))
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main()
