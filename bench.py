#!/usr/bin/env python3
"""bench.py -- full-tree Lk() throughput of the HIP likelihood engine on MI355X (BASELINE.json metric).

One "step" = one complete `Lk(NULL, tree)`: refresh every edge's transition matrices (device PMat from
the eigen system), the whole post-order partial-likelihood traversal (n-2 site-updates per pattern),
the root-edge reduction, and the scalar back on the host -- everything the reference's Lk(NULL) does
(src/lk.c:443-649) on inputs already resident in HBM.  Workload at N=1: BASELINE configs[1]
(100 taxa x 50 000 nt patterns, GTR+G4).  N>1 ranks shard patterns (weak scaling: 50 000 patterns per
GPU, no data-path collective, one RCCL all-reduce of the per-shard lnL per evaluation).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def cpu_baseline(wl, sample_patterns, reps):
    """Reference AVX path (oracle/_ref, the real PhyML objects) on a bounded sample, 1 host core.
    Falls back to the repo's CPU restatement (kind 'port') if the reference binary did not travel."""
    from phyml_amd import synth
    tree, st, cfg = wl["tree"], wl["states"], wl["cfg"]
    n_s = min(sample_patterns, st.shape[1])
    drv = os.path.join(ROOT, "oracle", "_ref", "phyml_ref_driver")
    if os.path.exists(drv):
        try:
            tmp = tempfile.mkdtemp(prefix="phyhip_cpu_")
            ali, tre = os.path.join(tmp, "a.phy"), os.path.join(tmp, "t.nwk")
            synth.write_phylip(ali, tree.names, synth.states_to_chars(st[:, :n_s], cfg["ns"]))
            open(tre, "w").write(tree.to_newick() + "\n")
            man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
            if cfg["ns"] == 4:
                margs, dopts = ["-d", "nt", "-m", "GTR", "-f", man["nt_freq"]], ["--gtr-rr", man["gtr_rr"]]
            else:
                margs, dopts = ["-d", "aa", "-m", "LG", "-f", "m"], []
            out = subprocess.run([drv, "bench", str(reps)] + dopts + ["--", "-i", ali, "-u", tre] + margs +
                                 ["-c", "4", "-a", "1.0", "-o", "n", "-b", "0", "--no_colalias"],
                                 cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600).stdout
            import re
            info = json.loads(re.search(r"REF_BENCH (\{.*\})", out).group(1))
            return dict(value=info["site_updates_per_s"] / 1e6, unit="M site-updates/s", cores=1, kind="reference",
                        sample=f"PhyML AVX Lk(NULL) x{reps} on {info['n_otu']} taxa x {info['n_pattern']} patterns of the same workload",
                        lnL_sample=info["lnL"])
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"[bench] reference baseline unavailable ({e}); using the CPU port\n")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    m = orc.Model(wl["model"])
    n_s = min(n_s, 4000)
    tv, ds, amb = [], [], []
    for t in range(tree.n_otu):
        v, s, a = orc.init_tip(m.datatype, synth.states_to_chars(st[t:t + 1, :n_s], cfg["ns"])[0])
        tv.append(v); ds.append(s); amb.append(a)
    ot = orc.OracleTree(m, tree.n_otu, tree.edge_left, tree.edge_rght, tree.edge_len, np.ones(n_s), tv, ds, amb)
    ot.lk(None)
    t0 = time.perf_counter()
    for _ in range(reps):
        lnl = ot.lk(None)
    dt = (time.perf_counter() - t0) / reps
    return dict(value=n_s * (tree.n_otu - 2) / dt / 1e6, unit="M site-updates/s", cores=1, kind="port",
                sample=f"oracle Lk(NULL) x{reps} on {tree.n_otu} taxa x {n_s} patterns", lnL_sample=lnl)


def measured_roofline():
    """HBM streaming rates of this GPU from the repo's micro-benchmark (phyml_amd/lib/membench): the
    denominator the north star calls the 'measured HBM-read roofline'."""
    mb = os.path.join(ROOT, "phyml_amd", "lib", "membench")
    try:
        out = subprocess.run([mb, "1", "7"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120).stdout
        return json.loads(out.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        return None


def pmc_traffic(workload):
    """HBM bytes per traversal launch from the committed rocprofv3 PMC passes of this same command
    (profiles/): 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes."""
    f = os.path.join(ROOT, "profiles", f"r01_pmc_{workload}.json")
    try:
        d = json.load(open(f))
        return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
    except Exception:  # noqa: BLE001
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2_nt_100x50k")
    ap.add_argument("--patterns", type=int, default=None, help="patterns per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=50000)
    ap.add_argument("--cpu-reps", type=int, default=30)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    force_dist = os.environ.get("PHYHIP_BENCH_FORCE_DIST") == "1"   # exercise the sharded path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)

    from phyml_amd import lktree, shard, workloads
    wl = workloads.make(args.workload, n_pattern=args.patterns,
                        seed=None if world == 1 else workloads.CONFIGS[args.workload]["seed"] + 1000 * rank)
    tree, st, blk, cfg = wl["tree"], wl["states"], wl["model"], wl["cfg"]
    n, P, S = tree.n_otu, st.shape[1], cfg["ns"]
    C = int(blk["ncatg"][0])

    t = lktree.LkTree(n, tree.edge_left, tree.edge_rght, tree.edge_len, P, S, C, device=local)
    t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"],
                float(blk["l_min"][0]), float(blk["l_max"][0]), 1.0, 1)
    t.Make_Tree_For_Lk(np.ones(P))
    t.set_tips(tip_states=st.astype(np.int32))
    dev_lnl = torch.zeros(2, dtype=torch.float64, device=f"cuda:{local}")
    stream = torch.cuda.current_stream()
    t.inst.set_stream(stream.cuda_stream)

    def step():
        if dist is None:
            return t.Lk(None)
        # sharded evaluation: per-shard lnL stays on the device, ONE all-reduce over RCCL, then the host reads it
        t.Lk_Shard_Device(dev_lnl.data_ptr())
        shard.allreduce_sum(dev_lnl, dist)
        return float(dev_lnl[0].item())

    lnl = None
    for _ in range(args.warmup):
        lnl = step()
    t.inst.profile(1)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lnl = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kern_ms, kern_n, kern_upd = t.inst.profile_read()

    out = None
    if rank == 0:
        updates_per_step = float(P) * (n - 2) * world
        value = updates_per_step * args.steps / dt / 1e6
        alg_bytes = workloads.algorithmic_bytes_per_pattern(n, S, C) * float(P)   # per launch (this rank)
        kdur = kern_ms / max(kern_n, 1) * 1e-3
        achieved = alg_bytes / kdur / 1e9 if kdur > 0 else 0.0
        out = {
            "metric": "M partial-lk site-updates/sec on full-tree Lk()", "value": value, "unit": "M site-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n} taxa x {P} {'nt' if S == 4 else 'aa'} patterns per GPU, "
                                   f"{'GTR' if S == 4 else 'LG'}+G{C}, fixed random tree, full post-order Lk(NULL) incl. P-matrix refresh and root-edge reduce",
                       "patterns_per_gpu": P, "taxa": n, "states": S, "rate_categories": C,
                       "parallelism": f"pattern-shard x{world}" if world > 1 else "single GPU"},
            "lnL": lnl,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": pmc_traffic(args.workload) if (world == 1 and args.patterns is None) else None,
                         "kernel": ("traverse_nt2_kernel" if C <= 4 else "traverse_nt_kernel") if S == 4 else "traverse_aa_kernel", "kernel_avg_us": kdur * 1e6,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "algorithmic_bytes_per_site_update": alg_bytes / (float(P) * (n - 2))},
        }
        if S == 20:
            # 20-state path: FP64 MFMA work issued per launch, per (tile, category, operation): rows 0..15 as ten
            # 16x16x4 (2048 flop) and rows 16..19 as ten four-block 4x4x4 (512 flop) -- all of it useful
            mfma_flops = float((P + 15) // 16) * C * (n - 2) * (10 * 2048.0 + 10 * 512.0)
            out["mfma"] = {"achieved": mfma_flops / kdur / 1e12, "peak": 78.6,
                           "unit": "TFLOP/s issued (f64 16x16x4 + 4x4x4_4b)",
                           "frac": mfma_flops / kdur / 1e12 / 78.6, "useful_frac_of_issued": 1.0}
        if world == 1:
            mr = measured_roofline()
            if mr:
                out["roofline"]["measured_read_GBps"] = mr["read_GBps"]
                out["roofline"]["measured_write_GBps"] = mr["write_GBps"]
                out["roofline"]["frac_of_measured_read"] = achieved / mr["read_GBps"]
        exp = workloads.manifest()["expected"].get(args.workload)
        if exp and world == 1 and P == exp["n_pattern"]:
            from phyml_amd import synth
            out["lnL_reference_avx"] = exp["lnL"]
            out["lnL_rel_err"] = abs(lnl - exp["lnL"]) / abs(exp["lnL"])
            out["input_checksum_ok"] = bool(synth.states_checksum(st) == exp["checksum"])
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_sample, args.cpu_reps)
    t.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes out last, after RCCL has finished any chatter of its own on stdout
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    # RCCL prints a version banner on stdout while the interpreter shuts down; leave before that so the JSON
    # line stays the last (and, single-GPU, the only) line on stdout
    sys.stdout.flush(); sys.stderr.flush()
    if dist is not None:
        os._exit(0)  # (not when single-process: profilers such as rocprofv3 flush their traces at normal exit)


if __name__ == "__main__":
    main()
