#!/usr/bin/env python3
"""tools/multigpu_selfcheck.py [--patterns P] [--devices 0,1,...] : first contact with more than one device, stage by stage.

With >= 2 devices visible (or a repeated device list, e.g. --devices 0,0, on a one-GPU box) builds ONE sharded instance of
cfg4's shape (100 taxa x P patterns), evaluates Lk(NULL) a few times and prints
  * lnL against the sum of the shards' own lnLs evaluated as plain single-device instances (1e-12 relative),
  * per stage, host-timed: all shards' launches issued (helper threads), shard kernels finished (stream sync per device),
    all-reduce + publish (the collective path alone, phyhip's own timer around reduce_and_publish is not exposed: measured as
    whole step minus a step of the slowest shard alone),
  * the whole step and the implied speed-up over one device holding all P patterns.
Every number it needs comes through the public C ABI; it changes nothing.  Developer tool for the first multi-GPU box."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patterns", type=int, default=1000000)
    ap.add_argument("--devices", default=None)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from phyml_amd import shard, workloads
    ndev = torch.cuda.device_count()
    devs = [int(x) for x in args.devices.split(",")] if args.devices else list(range(ndev))
    if len(devs) < 2:
        raise SystemExit("multigpu_selfcheck: one device visible -- pass --devices 0,0 to run the mechanism on it")
    name, P, G = "cfg4_nt_100x1M", args.patterns, len(devs)
    out = {"devices": devs, "patterns": P}

    def sync_all():
        for d in sorted(set(devs)):
            torch.cuda.synchronize(d)

    # every shard alone on its device: the numbers the sharded instance must reproduce, and the slowest shard's step
    shard_lnl, shard_ms = [], []
    for g, d in enumerate(devs):
        lo, hi = shard.shard_range(P, g, G)
        wl = workloads.make(name, n_pattern=hi - lo, pattern_offset=lo)
        t = bench.build_tree(wl, device=d)
        dt, lnl = bench.timed_steps(t, args.steps, 3, lambda: torch.cuda.synchronize(d))
        t.close()
        shard_lnl.append(lnl); shard_ms.append(dt / args.steps * 1e3)
    out["shard_lnL_sum"] = float(np.sum(shard_lnl)); out["shard_ms_alone"] = shard_ms
    # the sharded instance
    wl = workloads.make(name, n_pattern=P)
    t = bench.build_tree(wl, devices=devs)
    out["rccl_ranks"] = t.inst.comm_size()
    dt, lnl = bench.timed_steps(t, args.steps, 3, sync_all)
    out["sharded_ms_per_step"] = dt / args.steps * 1e3
    out["sharded_lnL"] = lnl
    out["lnL_rel_err_vs_shard_sum"] = abs(lnl - out["shard_lnL_sum"]) / abs(out["shard_lnL_sum"])
    out["collective_path_ms"] = out["sharded_ms_per_step"] - max(shard_ms)
    # launch skew: time for the call to return when nothing is waited for is not observable through the ABI; the step of a
    # one-shard "sharded" instance on device 0 against the plain instance isolates helper thread + collective on one device
    lo, hi = shard.shard_range(P, 0, G)
    wl0 = workloads.make(name, n_pattern=hi - lo, pattern_offset=lo)
    t0 = bench.build_tree(wl0, devices=[devs[0]])
    d0, _ = bench.timed_steps(t0, args.steps, 3, lambda: torch.cuda.synchronize(devs[0]))
    t0.close()
    out["one_shard_with_collective_ms"] = d0 / args.steps * 1e3
    # the search's call pattern on the sharded instance (src/spr.c:640-646, src/optimiz.c:607-663): short scalar-returning calls are
    # answered shard by shard (each shard's resident evaluators) and added by the host -- no launch, no collective per call
    from phyml_amd import replay
    tree = wl["tree"]
    t.Set_Both_Sides(True); t.Lk(None)
    tr = replay.make_trace(wl["states"].shape[0], tree.edge_left, tree.edge_rght, tree.edge_len, 400, seed=3,
                           walk_every=3, opt_every=4, n_dlk=5)
    t.Replay_Surface_Trace({k: v[:100] for k, v in tr.items()})
    c0 = time.perf_counter()
    vals, _ = t.Replay_Surface_Trace(tr)
    out["call_pattern_us_per_candidate"] = (time.perf_counter() - c0) / 400 * 1e6
    out["call_pattern_served_by_resident_workgroups"] = [int(t.inst.resident_stats(k)[0]) for k in (0, 1, 2)]
    out["call_pattern_finite"] = bool(np.isfinite(vals).all())
    out["call_pattern_route"] = ("every shard's resident evaluators, shard sums added on the host" if sum(out["call_pattern_served_by_resident_workgroups"]) > 0
                                 else "a launch per shard + the collective (large shards that share a device, or PHYHIP_SHARD_HOST_COMBINE=0)")
    t.close()
    if len(set(devs)) == len(devs):
        wl1 = workloads.make(name, n_pattern=P)
        t1 = bench.build_tree(wl1, device=devs[0])
        d1, l1 = bench.timed_steps(t1, args.steps, 3, lambda: torch.cuda.synchronize(devs[0]))
        t1.close()
        out["one_gpu_ms_per_step"] = d1 / args.steps * 1e3
        out["speedup"] = out["one_gpu_ms_per_step"] / out["sharded_ms_per_step"]
    print(json.dumps(out, indent=1))
    ok = out["lnL_rel_err_vs_shard_sum"] < 1e-12 and out["call_pattern_finite"]
    print("OK" if ok else "MISMATCH")
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0 if ok else 1)  # (RCCL prints a banner of its own on stdout while the interpreter shuts down)


if __name__ == "__main__":
    main()
