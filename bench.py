#!/usr/bin/env python3
"""bench.py -- full-tree Lk() throughput of the HIP likelihood engine on MI355X (BASELINE.json metric).

One "step" = one complete `Lk(NULL, tree)`: refresh every edge's transition matrices (device PMat from the eigen
system), the whole post-order partial-likelihood traversal (n-2 site-updates per pattern), the root-edge reduction and
the scalar back on the host -- everything the reference's Lk(NULL) does (src/lk.c:443-649), inputs resident in HBM.

  --gpus 1            BASELINE configs[1] (cfg2: 100 taxa x 50 000 nt patterns, GTR+G4) on one MI355X; the JSON line
                      also carries `roofline`, `cpu_baseline` (real PhyML AVX, 1 host core, same box, same run) and,
                      under `extra`, the 20-state configuration (cfg3: 200 taxa x 10 000 aa patterns, LG+G4) and
                      `call_latency`: microseconds per SPR regraft candidate / per scalar-returning call of a seeded
                      SPR + Br_Len_Opt call stream at the cfg5 size and at the reference's example-alignment size.
  --gpus N (N > 1)    BASELINE configs[3] (cfg4: 100 taxa x 1 000 000 nt patterns), STRONG scaling: contiguous pattern
                      shards of 1e6/N per GPU, ONE RCCL all-reduce of {warning, lnL} per evaluation inside libphyhip.so.
                      Launched by the driver as N ranks (torch.distributed.run, WORLD_SIZE = N): one process per GPU,
                      phyhip_comm_init_rank on an id broadcast through torch.distributed (gloo: rendezvous, barrier and the
                      MAX over the ranks' timers are host-side; the only RCCL communicator is the library's own).  Launched bare
                      (`python bench.py --gpus N`): ONE process drives all N devices through the sharded instance of
                      the C ABI (phyhip_create_instance with a resource list of N devices, ncclCommInitAll).
Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

METRIC = "M partial-lk site-updates/sec on full-tree Lk()"


def cpu_model():
    """The host CPU's model string and core count (what the one-core baseline ran on)."""
    model, cores = "unknown", os.cpu_count() or 0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, cores


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
            cpu, ncores = cpu_model()
            return dict(value=info["site_updates_per_s"] / 1e6, unit="M site-updates/s", cores=1, kind="reference", cpu=cpu, host_cores=ncores,
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
    cpu, ncores = cpu_model()
    return dict(value=n_s * (tree.n_otu - 2) / dt / 1e6, unit="M site-updates/s", cores=1, kind="port", cpu=cpu, host_cores=ncores,
                sample=f"oracle Lk(NULL) x{reps} on {tree.n_otu} taxa x {n_s} patterns", lnL_sample=lnl)


def measured_roofline():
    """HBM streaming rates of this GPU from the repo's micro-benchmark (phyml_amd/lib/membench): the
    denominator the north star calls the 'measured HBM-read roofline'."""
    mb = os.path.join(ROOT, "phyml_amd", "lib", "membench")
    try:
        out = subprocess.run([mb, "1", "7", "sweep"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=180).stdout
        return json.loads(out.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        return None


def kernel_source_hash():
    """Identity of the kernels a counter profile belongs to: hash of the device code, i.e. the kernel headers under
    phyml_amd/csrc (all kernels of the path live in *.hpp; the *.hip units and phyhip_host.hpp / phyhip_shard.hpp are the
    host side)."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "phyml_amd", "csrc")
    for f in sorted(os.listdir(base)):
        p = os.path.join(base, f)
        if os.path.isfile(p) and f.endswith(".hpp") and f not in ("phyhip_host.hpp", "phyhip_shard.hpp"):
            h.update(f.encode()); h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(workload):
    """HBM bytes per traversal launch from the rocprofv3 PMC passes of this same command (tools/profile_round.sh writes
    profiles/rNN_pmc_<workload>.json with the kernel-source hash it was taken on): 2 x FETCH_SIZE (gfx950 correction,
    MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes.  A profile taken on other kernel sources is refused
    (null): counters do not travel across kernel changes."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{workload}.json")), reverse=True):  # newest round first
        try:
            d = json.load(open(f))
            if d.get("kernel_source_hash") == kernel_source_hash():
                return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        except Exception:  # noqa: BLE001
            pass
    return None


VIRTUAL = None  # --virtual-buffers: threshold handed to phyhip_set_virtual_buffers (None: the library's default, 0: off)


def pmc_kernel(profile, kernel_substr):
    """HBM bytes per dispatch of one kernel from the counter profiles of the other call kinds (tools/profile_r05_extra.sh ->
    profiles/rNN_pmc_<profile>.json: per-kernel averages, 2 x FETCH_SIZE + WRITE_SIZE), or None when no profile of these very
    kernel sources is committed."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{profile}.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("kernel_source_hash") != kernel_source_hash():
                continue
            for k, v in d.get("kernels", {}).items():
                if kernel_substr in k and "hbm_bytes_per_dispatch" in v:
                    return float(v["hbm_bytes_per_dispatch"])
        except Exception:  # noqa: BLE001
            pass
    return None


def build_tree(wl, device=None, devices=None, virtual="default"):
    from phyml_amd import lktree
    tree, st, blk, cfg = wl["tree"], wl["states"], wl["model"], wl["cfg"]
    n, P, S = tree.n_otu, st.shape[1], cfg["ns"]
    C = int(blk["ncatg"][0])
    t = lktree.LkTree(n, tree.edge_left, tree.edge_rght, tree.edge_len, P, S, C, device=device, devices=devices)
    t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"],
                float(blk["l_min"][0]), float(blk["l_max"][0]), 1.0, 1)
    t.Make_Tree_For_Lk(np.ones(P))
    t.set_tips(tip_states=st.astype(np.int32))
    v = VIRTUAL if virtual == "default" else virtual
    if v is not None:
        t.inst.set_virtual_buffers(int(v))
    return t


def all_stored_companion(wl, args, torch, n, P):
    """The same workload with every buffer stored (phyhip_set_virtual_buffers(0)): what the traversal costs when the tip x tip
    results are written like all the others -- reported next to the headline so that nobody has to guess what the virtual
    buffers are worth."""
    t = build_tree(wl, device=0, virtual=0)
    steps = max(5, min(args.steps, 20))
    dt, lnl = timed_steps(t, steps, min(args.warmup, 5), torch.cuda.synchronize)
    kern_ms, kern_n, _ = t.inst.profile_read()
    _, wr = t.inst.profile_read_traffic()
    t.close()
    return {"ms_per_step": dt / steps * 1e3, "kernel_avg_us": kern_ms / max(kern_n, 1) * 1e3, "value": float(P) * (n - 2) * steps / dt / 1e6,
            "write_bytes": wr / max(kern_n, 1), "lnL": lnl, "steps": steps}


def materialise_after_traversal(wl, torch, reps=6):
    """What the buffers a whole-tree traversal left virtual cost the search that follows it: Lk(NULL) (both sides), then Lk(b) on an
    edge whose inner side is a tip x tip result -- the first SHORT launch that reads a virtual buffer, which stores them all
    (include/phyhip.h: phyhip_set_virtual_buffers) -- against the same call repeated (nothing virtual any more).  Per call, wall
    clock, the scalar on the host both times."""
    tree = wl["tree"]
    n = tree.n_otu
    deg_tips = {}
    for e, (a, b) in enumerate(zip(tree.edge_left, tree.edge_rght)):
        for u, v in ((int(a), int(b)), (int(b), int(a))):
            if u >= n and v < n:
                deg_tips[u] = deg_tips.get(u, 0) + 1
    cherry = next(u for u, k in sorted(deg_tips.items()) if k == 2)
    e_c = next(e for e, (a, b) in enumerate(zip(tree.edge_left, tree.edge_rght))
               if (int(a) == cherry and int(b) >= n) or (int(b) == cherry and int(a) >= n))
    t = build_tree(wl, device=0)
    t.Set_Both_Sides(True)
    first, again, virt = [], [], []
    for k in range(reps + 2):
        ref = t.Lk(None)
        torch.cuda.synchronize()
        v0 = t.inst.virtual_stats()
        t0 = time.perf_counter()
        a = t.Lk(e_c)
        t1 = time.perf_counter()
        b = t.Lk(e_c)
        t2 = time.perf_counter()
        v1 = t.inst.virtual_stats()
        # (a: a list-form launch -- the stored definitions in front -- adds its block sums in the two-wave-shape partition, b: a short
        # evaluation in the one-shape partition: the same tree state to the last bits, not necessarily the same double)
        assert abs(a - ref) <= 1e-9 * abs(ref) and abs(a - b) <= 1e-12 * abs(ref), (ref, a, b)
        if k >= 2:  # (the first rounds load code objects)
            first.append((t1 - t0) * 1e6); again.append((t2 - t1) * 1e6); virt.append((v0[0], v1[0], v1[3] - v0[3]))
    t.close()
    f, g = sum(first) / len(first), sum(again) / len(again)
    return {"materialise_after_traversal_us": f - g, "first_short_call_after_traversal_us": f, "same_call_repeated_us": g,
            "virtual_before_after_materialised": virt[-1], "edge": e_c, "both_sides": True, "reps": reps}


def timed_steps(t, steps, warmup, sync, barrier=None):
    lnl = None
    for _ in range(warmup):
        lnl = t.Lk(None)
    t.inst.profile(1)
    if barrier:
        barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        lnl = t.Lk(None)
    sync()
    if barrier:
        barrier()
    return time.perf_counter() - t0, lnl


def roofline_block(t, n, P, S, C, workload_key, with_traffic, shards=1):
    """SURVEY 8(d) figure (algorithmic bytes / kernel time) plus the honest companions: the traffic model of the launch
    (results written once, children read unless forwarded in registers or tips), the write stream alone, and -- when a
    counter profile of these very kernel sources is committed -- the HBM bytes the counters saw."""
    from phyml_amd import workloads
    kern_ms, kern_n, _ = t.inst.profile_read()
    rd, wr = t.inst.profile_read_traffic()
    kdur = kern_ms / max(kern_n, 1) * 1e-3
    alg_bytes = workloads.algorithmic_bytes_per_pattern(n, S, C) * float(P)
    achieved = alg_bytes / kdur / 1e9 if kdur > 0 else 0.0
    rd, wr = rd / max(kern_n, 1) / shards, wr / max(kern_n, 1) / shards  # per launch of ONE shard
    pmc = pmc_traffic(workload_key) if with_traffic else None
    # `traffic`: the HBM bytes of one launch -- the counters' (2 x FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 --pmc passes of
    # this very command) when a profile of these kernel sources is committed, else the launch's own count of what it must move
    # (phyhip_profile_read_traffic: results stored once, children read unless forwarded in registers; the counters read 1.03 x
    # that in round 5).  `frac` = traffic / kernel time / 8 TB/s: a PHYSICAL fraction, never above 1.  SURVEY 8(d)'s algorithmic
    # bytes (every child read and result write of a post-order, also the ones a fused launch keeps in registers) are kept as
    # `frac_algorithmic`: that figure is not a rate of the memory system and exceeds 1 at large sizes.
    traffic = pmc if pmc else rd + wr
    vnow, vskip, vre, vstored = t.inst.virtual_stats()
    phys = traffic / kdur / 1e9 if kdur > 0 else 0.0
    r = {"bound": "hbm", "achieved": phys, "peak": 8000.0, "unit": "GB/s", "frac": phys / 8000.0, "traffic": traffic,
         "traffic_source": "rocprofv3 pmc (2*FETCH_SIZE+WRITE_SIZE, KiB)" if pmc else "launch byte model (no pmc profile of these kernel sources)",
         "frac_is": "physical: traffic / kernel time / 8 TB/s",
         "frac_algorithmic": achieved / 8000.0, "achieved_algorithmic": achieved,
         "frac_real": (pmc / kdur / 8e12) if (pmc and kdur > 0) else None,
         # buffers whose tip x tip result the launch did NOT store (include/phyhip.h: phyhip_set_virtual_buffers) -- they are
         # recomputed in registers in front of their readers and stored on demand; `all_buffers_stored` is the same run without that
         "virtual_buffers": {"virtual_after_launch": vnow, "internal_buffers": n - 2, "stores_skipped": vskip,
                             "reissued_not_storing": vre, "materialised": vstored},
         "kernel": t.inst.profile_read_kernel(),
         "kernel_avg_us": kdur * 1e6, "algorithmic_bytes_per_launch": alg_bytes,
         "algorithmic_bytes_per_site_update": alg_bytes / (float(P) * (n - 2)),
         "min_traffic_bytes": rd + wr, "min_read_bytes": rd, "write_bytes": wr,
         "frac_of_min_traffic": (rd + wr) / kdur / 8e12 if kdur > 0 else 0.0,
         "frac_of_write_stream": wr / kdur / 8e12 if kdur > 0 else 0.0}
    if pmc:
        r["hbm_achieved"] = pmc / kdur / 1e9
        r["hbm_frac"] = pmc / kdur / 8e12
    return r, kdur


def run_single(args, torch):
    """N = 1: cfg2 on cuda:0 (the line BENCH records), cfg3 under `extra`."""
    from phyml_amd import synth, workloads
    torch.cuda.set_device(0)
    wl = workloads.make(args.workload, n_pattern=args.patterns)
    tree, st, cfg = wl["tree"], wl["states"], wl["cfg"]
    n, P, S, C = tree.n_otu, st.shape[1], cfg["ns"], int(wl["model"]["ncatg"][0])
    t = build_tree(wl, device=0)
    dt, lnl = timed_steps(t, args.steps, args.warmup, torch.cuda.synchronize)
    value = float(P) * (n - 2) * args.steps / dt / 1e6
    roof, kdur = roofline_block(t, n, P, S, C, args.workload, args.patterns is None)
    if roof["virtual_buffers"]["virtual_after_launch"] > 0 and not args.no_companion:
        roof["all_buffers_stored"] = all_stored_companion(wl, args, torch, n, P)
        try:
            roof["materialise"] = materialise_after_traversal(wl, torch)
        except Exception as e:  # (a companion must never cost the run its headline line)
            roof["materialise"] = {"error": repr(e)}
    # ("scaling": the N = 1 line is the cfg2 workload on one GPU -- neither weak nor strong; --gpus N > 1 is STRONG scaling of cfg4)
    out = {"metric": METRIC, "value": value, "unit": "M site-updates/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "none", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": f"{args.workload}: {n} taxa x {P} {'nt' if S == 4 else 'aa'} patterns, "
                                  f"{'GTR' if S == 4 else 'LG'}+G{C}, fixed random tree, full post-order Lk(NULL) incl. P-matrix refresh and root-edge reduce",
                      "patterns_per_gpu": P, "taxa": n, "states": S, "rate_categories": C, "parallelism": "single GPU"},
           "lnL": lnl, "roofline": roof}
    if S == 20:
        out["mfma"] = mfma_block(P, C, n, kdur)
    mr = measured_roofline()
    if mr:
        roof["measured_read_GBps"] = mr["read_GBps"]
        roof["measured_write_GBps"] = mr["write_GBps"]
        # the streams this GPU sustains (phyml_amd/lib/membench: plain and NON-TEMPORAL accesses -- the traversal kernels' policy --
        # over 1-32 workgroups per CU of 256 / 64 lanes; best_* = the best geometry): what "fraction of the measured roofline" divides by
        for k in ("read_nt_GBps", "write_nt_GBps", "r1w2_GBps", "r1w2_nt_GBps", "best_read_GBps", "best_write_GBps", "best_copy_GBps",
                  "best_r1w2_GBps"):
            if k in mr:
                roof["measured_" + k] = mr[k]
        mixed = max(mr.get("best_r1w2_GBps", 0.0), mr.get("r1w2_nt_GBps", 0.0), mr.get("r1w2_GBps", 0.0))
        best_write = max(mr.get("best_write_GBps", 0.0), mr.get("write_nt_GBps", 0.0), mr["write_GBps"])
        # SURVEY 8(d)'s unit over the measured read stream.  The algorithmic bytes charge every child of every operation; a
        # fused launch forwards most children in registers, so this can exceed 1 -- it is NOT a physical rate (those follow)
        roof["algorithmic_frac_of_measured_read"] = roof["achieved_algorithmic"] / max(mr["read_GBps"], mr.get("best_read_GBps", 0.0))
        roof["write_stream_frac_of_measured_write"] = roof["write_bytes"] / kdur / 1e9 / best_write
        if roof.get("hbm_achieved") and mixed > 0:
            roof["frac_real_bytes_of_measured_mixed"] = roof["hbm_achieved"] / mixed
        elif mixed > 0:  # (no counter profile of these kernel sources: the launch's traffic model instead)
            roof["frac_min_traffic_of_measured_mixed"] = roof["min_traffic_bytes"] / kdur / 1e9 / mixed
    exp = workloads.manifest()["expected"].get(args.workload)
    if exp and P == exp["n_pattern"]:
        out["lnL_reference_avx"] = exp["lnL"]
        out["lnL_rel_err"] = abs(lnl - exp["lnL"]) / abs(exp["lnL"])
        if "checksum" in exp:
            out["input_checksum_ok"] = bool(synth.states_checksum(st) == exp["checksum"])
    t.close()
    if not args.no_extra and args.workload == "cfg2_nt_100x50k" and args.patterns is None:
        out["extra"] = {"cfg3_aa_200x10k": extra_line("cfg3_aa_200x10k", args, torch)}
        # the strong-scaling reference of the N > 1 lines (cfg4, 100 taxa x 1 M patterns) on THIS one GPU, same protocol: a
        # 1 -> 8 curve over one workload is value(N > 1) against this value, whatever the N = 1 headline workload is
        try:
            out["extra"]["cfg4_nt_100x1M_one_gpu"] = scaling_reference_line(args, torch)
        except Exception as e:  # (an extra must never cost the run its headline line)
            out["extra"]["cfg4_nt_100x1M_one_gpu"] = {"error": repr(e)}
        if not args.no_call_latency:
            try:
                out["extra"]["call_latency"] = call_latency()
            except Exception as e:  # (an extra must never cost the run its headline line)
                out["extra"]["call_latency"] = {"error": repr(e)}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, args.cpu_sample, args.cpu_reps)
    return out


def mfma_block(P, C, n, kdur):
    # 20-state path: FP64 MFMA work issued per launch.  A wave-tile (16 / blocks-per-pattern patterns x all categories)
    # spends 25 v_mfma_f64_4x4x4_4b (512 flop each) per INTERNAL child of an operation -- the n - 1 one-state tip children of
    # this synthetic alignment read a matrix column instead (phyhip_aa.hpp) -- and 25 for the evaluation edge
    cb = 1 if C == 1 else (2 if C == 2 else 4)
    tiles = float((P + 15) // 16 * 16) / (16 // cb)
    flops = tiles * ((n - 3) + 1) * 25 * 512.0
    return {"achieved": flops / kdur / 1e12, "peak": 78.6, "unit": "TFLOP/s issued (f64 4x4x4_4b)",
            "frac": flops / kdur / 1e12 / 78.6, "useful_frac_of_issued": C / float(cb),
            "note": "one-state tip children cost no matrix-core work (column lookup): half of the products of a full post-order"}


def call_latency():
    """The launch-bound side of the path, driver-timed in the same run: the seeded SPR / Br_Len_Opt call stream of
    phyml_amd/replay.py (a regraft candidate = 3 matrix refreshes + 1 partial update + the edge likelihood with the scalar on
    the host, src/spr.c:643-646; every `opt` candidates a branch-length chain of dLk calls, src/optimiz.c:607-663) at the
    cfg5 size and at the size of the reference's own example alignment.  Every scalar of this stream is checked
    against the oracle in tests/test_gpu_cfg5.py / test_gpu_replay.py / test_gpu_resident.py; here it is only timed."""
    import numpy as np
    from phyml_amd import lktree, replay, synth, workloads
    rows = {}
    # (the 20-state rows: the size of the reference's example protein alignment and cfg3's -- a launch per call, there is no
    # resident evaluator for 20 states)
    for name, taxa, P, cand, opt, S in (("spr_500x100k", 500, 100000, 600, 0, 4), ("spr_54x382", 54, 382, 3000, 0, 4),
                                        ("spr_and_brlen_54x382", 54, 382, 1500, 4, 4), ("spr_37x429_aa", 37, 429, 1500, 0, 20),
                                        ("spr_and_brlen_37x429_aa", 37, 429, 1000, 4, 20), ("spr_200x10k_aa", 200, 10000, 600, 0, 20)):
        blk = workloads.model_block("model_gtr_g4" if S == 4 else "model_lg_g4")
        tree = synth.random_tree(taxa, 9, 0.02, 0.15)
        st = synth.simulate_states(tree, P, S, 9)
        t = lktree.LkTree(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, P, S, int(blk["ncatg"][0]))
        t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"])
        t.Make_Tree_For_Lk(np.ones(P))
        t.set_tips(tip_states=st.astype(np.int32))
        t.Set_Both_Sides(True)
        t.Lk(None)
        tr = replay.make_trace(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, cand, seed=3, walk_every=3, opt_every=opt, n_dlk=5)
        t.Replay_Surface_Trace({k: v[:300] for k, v in tr.items()})  # warm-up (first launches load code)
        t0 = time.perf_counter()
        res, _ = t.Replay_Surface_Trace(tr)
        dt = time.perf_counter() - t0
        k = tr["kind"]
        n_scalar = int(np.isin(k, (replay.EDGE_LNL, replay.DLK)).sum())
        if name == "spr_500x100k":
            brlen = brlen_block(t, taxa, P, 4, int(blk["ncatg"][0]))
        extra = {}
        if name == "spr_500x100k":
            # what a candidate moves (counter profile of the large-grid kernel's one-shot form, the tile bodies the resident
            # workgroups run: profiles/rNN_pmc_cfg5_spr.json) against what it takes end to end -- latency, not a streaming rate
            b = pmc_kernel("cfg5_spr", "resident_big_kernel")
            if b:
                extra = {"hbm_bytes_per_candidate": b, "hbm_GBps_end_to_end": b / (dt / cand) / 1e9, "frac_real_end_to_end": b / (dt / cand) / 8e12}
        rows[name] = {**extra, "us_per_candidate": dt / cand * 1e6, "us_per_scalar_returning_call": dt / n_scalar * 1e6, "candidates": cand,
                      "scalar_returning_calls": n_scalar, "dlk_calls": int((k == replay.DLK).sum()), "surface_calls": int(len(k)),
                      "finite": bool(np.isfinite(res).all()),
                      "served_by_resident_workgroups": {"dlk": t.inst.resident_stats(0)[0], "short_evaluations": t.inst.resident_stats(1)[0],
                                                        "large_grid": t.inst.resident_stats(2)[0]}}
        t.close()
    rows["brlen_500x100k"] = brlen
    return rows


def brlen_block(t, taxa, P, S, C):
    """K3 / K4 at the cfg5 size: Update_Eigen_Lr's kernel streams two partial vectors in and the eigen-basis
    products out (3 x P C S 8 B), the dLk kernel streams the products in (P C S 8 B + weights + scale exponents); HIP events
    around each launch (phyhip_profile_read_eigen), Br_Len_Opt's call pattern: one Update_Eigen_Lr, then a chain of dLk."""
    e = t.ne // 2
    t.Lk(e)
    t.Update_Eigen_Lr(e)
    t.Set_Update_Eigen_Lr(0); t.Set_Use_Eigen_Lr(1)
    for i in range(3):
        t.dLk(0.05, e)
    # kernel times of K3 / K4 (HIP events around each launch: profiling an instance makes every call a launch) ...
    n_chain, n_dlk = 40, 5
    t.inst.profile(1)  # (one round unmeasured: the first launch of a kernel also loads its code object -- each translation unit's lazily)
    t.Set_Update_Eigen_Lr(1); t.Update_Eigen_Lr(e); t.Set_Update_Eigen_Lr(0)
    t.dLk(0.05, e)
    t.inst.profile(1)  # (resets the sums)
    for k in range(8):
        t.Set_Update_Eigen_Lr(1); t.Update_Eigen_Lr(e); t.Set_Update_Eigen_Lr(0)
        for i in range(n_dlk):
            t.dLk(0.05 + 1e-3 * i + 1e-5 * k, e)
    (k3_ms, k3_n), (k4_ms, k4_n) = t.inst.profile_read_eigen()
    t.inst.profile(0)
    # ... and the wall time of Br_Len_Opt's chain as a caller sees it (resident workgroups where the engine uses them)
    for k in range(4):
        t.Set_Update_Eigen_Lr(1); t.Update_Eigen_Lr(e); t.Set_Update_Eigen_Lr(0)
        for i in range(n_dlk):
            t.dLk(0.05 + 1e-3 * i, e)
    t0 = time.perf_counter()
    for k in range(n_chain):
        t.Set_Update_Eigen_Lr(1); t.Update_Eigen_Lr(e); t.Set_Update_Eigen_Lr(0)
        for i in range(n_dlk):
            t.dLk(0.05 + 1e-3 * i + 1e-5 * k, e)
    wall = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in range(200):
        t.dLk(0.05 + 1e-4 * i, e)
    dlk_wall = (time.perf_counter() - t0) / 200 * 1e6
    t.Set_Use_Eigen_Lr(0)
    vec = float(P) * C * S * 8.0
    k3_bytes, k4_bytes = 3.0 * vec + 2.0 * 4.0 * P, vec + 8.0 * P + 4.0 * P
    k3_us, k4_us = k3_ms / max(k3_n, 1) * 1e3, k4_ms / max(k4_n, 1) * 1e3
    return {"patterns": P, "taxa": taxa,
            "eigen_lr_kernel": {"launches": k3_n, "avg_us": k3_us, "bytes": k3_bytes, "GBps": k3_bytes / (k3_us * 1e-6) / 1e9 if k3_us else 0.0,
                                "frac_of_8TBps": k3_bytes / (k3_us * 1e-6) / 8e12 if k3_us else 0.0},
            "dlk_kernel": {"launches": k4_n, "avg_us": k4_us, "bytes": k4_bytes, "GBps": k4_bytes / (k4_us * 1e-6) / 1e9 if k4_us else 0.0,
                           "frac_of_8TBps": k4_bytes / (k4_us * 1e-6) / 8e12 if k4_us else 0.0},
            "us_per_chain_of_1_eigen_lr_and_5_dlk": wall / n_chain * 1e6, "us_per_dlk_call": dlk_wall,
            "hbm_bytes_per_dlk_call": pmc_kernel("cfg5_dlk", "resident_big_kernel"), "hbm_bytes_per_eigen_lr_kernel": pmc_kernel("cfg5_eig", "eigen_lr_kernel"),
            "served_by_large_grid_resident_workgroups": t.inst.resident_stats(2)[0]}


def scaling_reference_line(args, torch):
    """cfg4 (the workload of the N > 1 lines) whole on one GPU: the denominator of the strong-scaling curve."""
    from phyml_amd import workloads
    name = "cfg4_nt_100x1M"
    wl = workloads.make(name)
    tree, st = wl["tree"], wl["states"]
    n, P = tree.n_otu, st.shape[1]
    t = build_tree(wl, device=0)
    steps = max(5, min(args.steps, 20))
    dt, lnl = timed_steps(t, steps, max(3, min(args.warmup, 10)), torch.cuda.synchronize)
    kern_ms, kern_n, _ = t.inst.profile_read()
    rd_, wr_ = t.inst.profile_read_traffic()
    model_traffic = (rd_ + wr_) / max(kern_n, 1)
    kname = t.inst.profile_read_kernel()
    vnow = t.inst.virtual_stats()[0]
    t.close()
    exp = workloads.manifest()["expected"][name]
    kdur = kern_ms / max(kern_n, 1) * 1e-3
    alg = workloads.algorithmic_bytes_per_pattern(n, 4, 4) * float(P)
    pmc = pmc_kernel("cfg4_1M", "traverse_nt2")
    traffic = pmc if pmc else model_traffic
    roof = {"bound": "hbm", "kernel": kname, "kernel_avg_us": kdur * 1e6, "achieved": traffic / kdur / 1e9 if kdur > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
            "frac": traffic / kdur / 8e12 if kdur > 0 else 0.0, "frac_is": "physical: traffic / kernel time / 8 TB/s",
            "traffic": traffic, "traffic_source": "rocprofv3 pmc" if pmc else "launch byte model",
            "frac_algorithmic": alg / kdur / 8e12 if kdur > 0 else 0.0, "virtual_buffers_after_launch": vnow}
    return {"value": float(P) * (n - 2) * steps / dt / 1e6, "unit": "M site-updates/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "n_gpus": 1, "patterns": P, "lnL": lnl, "lnL_rel_err": abs(lnl - exp["lnL"]) / abs(exp["lnL"]), "roofline": roof,
            "note": "strong-scaling reference: bench.py --gpus N (N > 1) runs this workload in N pattern shards"}


def extra_line(name, args, torch):
    """A second, driver-timed configuration reported under `extra` (same protocol, same run)."""
    from phyml_amd import synth, workloads
    wl = workloads.make(name)
    tree, st, cfg = wl["tree"], wl["states"], wl["cfg"]
    n, P, S, C = tree.n_otu, st.shape[1], cfg["ns"], int(wl["model"]["ncatg"][0])
    t = build_tree(wl, device=0)
    dt, lnl = timed_steps(t, args.steps, args.warmup, torch.cuda.synchronize)
    roof, kdur = roofline_block(t, n, P, S, C, name, True)
    if roof["virtual_buffers"]["virtual_after_launch"] > 0 and not args.no_companion:
        roof["all_buffers_stored"] = all_stored_companion(wl, args, torch, n, P)
    exp = workloads.manifest()["expected"][name]
    line = {"value": float(P) * (n - 2) * args.steps / dt / 1e6, "unit": "M site-updates/s", "ms_per_step": dt / args.steps * 1e3,
            "steps": args.steps, "lnL": lnl, "lnL_rel_err": abs(lnl - exp["lnL"]) / abs(exp["lnL"]),
            "input_checksum_ok": bool(synth.states_checksum(st) == exp["checksum"]), "roofline": roof}
    if S == 20:
        line["mfma"] = mfma_block(P, C, n, kdur)
    t.close()
    return line


def run_sharded(args, torch):
    """N > 1: cfg4 (100 taxa x 1 M nt patterns), strong scaling, the all-reduce inside libphyhip.so."""
    from phyml_amd import capi, shard, workloads
    name = "cfg4_nt_100x1M"
    total = workloads.CONFIGS[name]["n_pattern"] if args.patterns is None else int(args.patterns)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    multiproc = world > 1 or os.environ.get("PHYHIP_BENCH_FORCE_DIST") == "1"
    exp = workloads.manifest()["expected"].get(name)
    if multiproc:
        if world != args.gpus and not (world == 1 and os.environ.get("PHYHIP_BENCH_FORCE_DIST") == "1"):
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        # rendezvous over gloo (CPU): the ONLY RCCL communicator on a device is the one libphyhip.so builds below
        dist.init_process_group("gloo")
        lo, hi = shard.shard_range(total, rank, world)
        wl = workloads.make(name, n_pattern=hi - lo, pattern_offset=lo)
        t = build_tree(wl, device=local)
        # the library's own communicator: id from rank 0, broadcast through torch.distributed (as an MPI host would MPI_Bcast it)
        ids = [capi.comm_get_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        t.inst.comm_init_rank(world, rank, ids[0])
        ranks_seen = t.inst.comm_size()
        barrier = dist.barrier
        n_gpus, mode = world, "one process per GPU: phyhip_comm_init_rank + ncclAllReduce inside libphyhip.so"
    else:
        ndev = torch.cuda.device_count()
        devs = list(range(args.gpus))
        if os.environ.get("PHYHIP_BENCH_DEVICES"):  # e.g. "0,0": exercise this path on a box with fewer devices than shards
            devs = [int(x) for x in os.environ["PHYHIP_BENCH_DEVICES"].split(",")]
            assert len(devs) == args.gpus
        if ndev <= max(devs):
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} device(s) visible")
        wl = workloads.make(name, n_pattern=total)
        t = build_tree(wl, devices=devs)
        ranks_seen = t.inst.comm_size()
        barrier = None
        n_gpus, mode = args.gpus, "one process, sharded instance: ncclCommInitAll + ncclAllReduce inside libphyhip.so"

    def sync():
        for d in range(torch.cuda.device_count() if not multiproc else 1):
            torch.cuda.synchronize(local if multiproc else d)

    tree, cfg = wl["tree"], wl["cfg"]
    n, S, C = tree.n_otu, cfg["ns"], int(wl["model"]["ncatg"][0])
    dt, lnl = timed_steps(t, args.steps, args.warmup, sync, barrier)
    if multiproc:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    P_local = wl["states"].shape[1] if multiproc else total // n_gpus
    roof, kdur = roofline_block(t, n, P_local, S, C, name, False, shards=1 if multiproc else n_gpus)
    # what every rank saw: its traversal kernel (HIP events), the collective path behind it (local sum, ncclAllReduce, publish:
    # HIP events on the rank's stream -- includes waiting for the slowest rank) and the size of the communicator RCCL built
    coll_ms, coll_n, coll_ranks = t.inst.profile_read_collective()
    mine = {"rank": rank, "kernel_us": round(kdur * 1e6, 2), "collective_us": round(coll_ms / max(coll_n, 1) * 1e3, 2), "comm_size": coll_ranks,
            "patterns": int(P_local)}
    if multiproc:
        every = [None] * world
        dist.all_gather_object(every, mine)
    else:
        every = [mine]  # (one process, sharded instance: the slowest shard's kernel, the first device's collective path)
    per_rank = {"kernel_us": [e["kernel_us"] for e in every], "collective_us": [e["collective_us"] for e in every],
                "comm_size": [e["comm_size"] for e in every], "patterns": [e["patterns"] for e in every]}
    t.close()
    out = None
    if rank == 0:
        value = float(total) * (n - 2) * args.steps / dt / 1e6
        out = {"metric": METRIC, "value": value, "unit": "M site-updates/s", "n_gpus": n_gpus, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{name}: {n} taxa x {total} nt patterns in {n_gpus} contiguous shards of {total // n_gpus}, GTR+G{C}, "
                                      "fixed random tree, full post-order Lk(NULL) incl. P-matrix refresh, root-edge reduce and ONE RCCL "
                                      "all-reduce of {warning, lnL} per evaluation",
                          "patterns_total": total, "patterns_per_gpu": total // n_gpus, "taxa": n, "states": S, "rate_categories": C,
                          "parallelism": f"pattern-shard x{n_gpus}", "mode": mode, "rccl_ranks": ranks_seen},
               "lnL": lnl, "roofline": roof, "per_rank": per_rank}
        roof["note"] = "per-GPU figures of rank 0's shard"
        if exp and total == exp["n_pattern"]:
            out["lnL_reference_avx"] = exp["lnL"]
            out["lnL_rel_err"] = abs(lnl - exp["lnL"]) / abs(exp["lnL"])
        if not args.no_extra:
            # the strong-scaling reference: the same 1 M patterns on ONE of these GPUs, same protocol, same run
            wl1 = workloads.make(name, n_pattern=total)
            t1 = build_tree(wl1, device=local if multiproc else 0)
            dt1, lnl1 = timed_steps(t1, args.steps, args.warmup, sync)
            t1.close()
            v1 = float(total) * (n - 2) * args.steps / dt1 / 1e6
            out["strong_scaling"] = {"single_gpu_value": v1, "single_gpu_ms_per_step": dt1 / args.steps * 1e3,
                                     "single_gpu_lnL": lnl1, "speedup": value / v1, "efficiency": value / v1 / n_gpus}
    if multiproc:
        dist.barrier()
        dist.destroy_process_group()
    return out, True   # RCCL was used in this process (either form): see the exit note in main()


def _r(x, n=4):
    if x is None or isinstance(x, (str, bool, int)):
        return x
    x = float(x)
    return round(x, n) if abs(x) < 1e6 else float(f"{x:.7g}")


def _roof_short(roof):
    """The roofline block of the printed line: the contract's keys (physical figures) + kernel name and time, the algorithmic
    companion, and -- where buffers stayed virtual -- the all-stored run and what materialising them costs the next short call."""
    o = {k: _r(roof.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    o["traffic_is"] = "pmc" if str(roof.get("traffic_source", "")).startswith("rocprofv3") else "model"
    o["kernel"] = roof.get("kernel")
    o["kernel_avg_us"] = _r(roof.get("kernel_avg_us"), 2)
    o["frac_algorithmic"] = _r(roof.get("frac_algorithmic"))
    vb = roof.get("virtual_buffers") or {}
    if "virtual_after_launch" in vb:
        o["virtual_buffers"] = vb["virtual_after_launch"]
    elif "virtual_buffers_after_launch" in roof:
        o["virtual_buffers"] = roof["virtual_buffers_after_launch"]
    a = roof.get("all_buffers_stored")
    if a:
        o["all_stored_kernel_us"] = _r(a["kernel_avg_us"], 2)
        o["all_stored_value"] = _r(a["value"], 1)
    m = roof.get("materialise")
    if m and "materialise_after_traversal_us" in m:
        o["materialise_after_traversal_us"] = _r(m["materialise_after_traversal_us"], 2)
    return o


def compact_line(out):
    """The printed line: the contract's keys, `roofline` and `cpu_baseline` in short, one short block per further configuration."""
    line = {k: out[k] for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value"] = _r(out["value"], 1)
    line["ms_per_step"] = _r(out["ms_per_step"], 5)
    cfg = out["config"]
    line["config"] = {"workload": cfg["workload"].split(":")[0], **{k: cfg[k] for k in ("taxa", "states", "rate_categories", "patterns_per_gpu", "patterns_total", "parallelism", "rccl_ranks") if k in cfg}}
    if "lnL_rel_err" in out:
        line["lnL_rel_err"] = float(f"{out['lnL_rel_err']:.2g}")
    line["roofline"] = _roof_short(out["roofline"])
    if "cpu_baseline" in out:
        c = out["cpu_baseline"]
        line["cpu_baseline"] = {"value": _r(c["value"], 2), "unit": c["unit"], "cores": c["cores"], "kind": c["kind"],
                                "sample": c["sample"].replace(" of the same workload", "")}
    ex = out.get("extra", {})
    c3 = ex.get("cfg3_aa_200x10k")
    if c3 and "value" in c3:
        r3 = _roof_short(c3["roofline"])
        line["cfg3_aa_200x10k"] = {"value": _r(c3["value"], 1), "ms_per_step": _r(c3["ms_per_step"], 5), "lnL_rel_err": float(f"{c3['lnL_rel_err']:.2g}"),
                                   **{k: r3[k] for k in ("kernel", "kernel_avg_us", "frac", "traffic", "traffic_is", "frac_algorithmic", "all_stored_kernel_us") if k in r3},
                                   "mfma_frac": _r(c3["mfma"]["frac"])}
    c4 = ex.get("cfg4_nt_100x1M_one_gpu")
    if c4 and "value" in c4:
        r4 = _roof_short(c4["roofline"])
        line["cfg4_nt_100x1M_one_gpu"] = {"value": _r(c4["value"], 1), "ms_per_step": _r(c4["ms_per_step"], 4), "lnL_rel_err": float(f"{c4['lnL_rel_err']:.2g}"),
                                          **{k: r4[k] for k in ("kernel_avg_us", "frac", "traffic_is", "frac_algorithmic") if k in r4}}
    cl = ex.get("call_latency")
    if cl and "error" not in cl:
        us = {}
        for name, short in (("spr_500x100k", "spr_500x100k"), ("spr_54x382", "spr_54x382"), ("spr_37x429_aa", "spr_37x429_aa"), ("spr_200x10k_aa", "spr_200x10k_aa")):
            if name in cl:
                us[short] = _r(cl[name]["us_per_candidate"], 2)
        for name, short in (("spr_and_brlen_54x382", "call_54x382"), ("spr_and_brlen_37x429_aa", "call_37x429_aa")):
            if name in cl:
                us[short] = _r(cl[name]["us_per_scalar_returning_call"], 2)
        b = cl.get("brlen_500x100k")
        if b:
            us["dlk_500x100k"] = _r(b["us_per_dlk_call"], 2)
            us["chain_1eig_5dlk_500x100k"] = _r(b["us_per_chain_of_1_eigen_lr_and_5_dlk"], 2)
        aa = [cl[k]["served_by_resident_workgroups"] for k in cl if k.endswith("_aa") and "served_by_resident_workgroups" in cl[k]]
        us["resident_served_aa"] = int(sum(sum(v.values()) for v in aa))
        line["call_us"] = us
    if "strong_scaling" in out:
        ss = out["strong_scaling"]
        line["strong_scaling"] = {"single_gpu_value": _r(ss["single_gpu_value"], 1), "speedup": _r(ss["speedup"], 3)}
    if "per_rank" in out:
        line["per_rank"] = out["per_rank"]
    line["detail"] = "bench_detail.json"
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2_nt_100x50k", help="N = 1 only")
    ap.add_argument("--patterns", type=int, default=None, help="pattern count override (N = 1: of the workload; N > 1: total)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-call-latency", action="store_true",
                    help="skip the call_latency part of `extra` (kernel-stats profiles: its launches share kernel names with the bench's)")
    ap.add_argument("--no-extra", action="store_true", help="skip the cfg3 line (N = 1) / the single-GPU reference (N > 1)")
    ap.add_argument("--virtual-buffers", type=int, default=None,
                    help="phyhip_set_virtual_buffers threshold (default: the library's, 16 operations; 0: every buffer stored)")
    ap.add_argument("--no-companion", action="store_true",
                    help="skip the all-buffers-stored companion run (counter profiles: its launches share the traversal kernel's name)")
    ap.add_argument("--cpu-sample", type=int, default=50000)
    ap.add_argument("--cpu-reps", type=int, default=30)
    args = ap.parse_args()
    global VIRTUAL
    VIRTUAL = args.virtual_buffers

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    used_rccl = False
    if args.gpus == 1 and os.environ.get("PHYHIP_BENCH_FORCE_DIST") != "1":
        out = run_single(args, torch)
    else:
        out, used_rccl = run_sharded(args, torch)
    if out is not None:
        # Everything measured goes to a side file (and stderr); the ONE JSON line on stdout is the compact form of it -- the
        # contract's keys plus the few figures a reader needs, under 2 000 characters so that a log tail holds all of it
        detail = json.dumps(out)
        for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
            try:
                if os.path.isdir(d):
                    open(os.path.join(d, "bench_detail.json"), "w").write(detail + "\n")
            except OSError:
                pass
        sys.stderr.write("[bench detail] " + detail + "\n")
        line = json.dumps(compact_line(out), separators=(",", ":"))
        assert len(line) < 2000, len(line)
        # the ONE JSON line goes out last, after RCCL has finished any chatter of its own on stdout
        sys.stdout.flush()
        print(line, flush=True)
    # RCCL prints a version banner on stdout while the interpreter shuts down; leave before that so the JSON
    # line stays the last (and, single-GPU, the only) line on stdout
    sys.stdout.flush(); sys.stderr.flush()
    if used_rccl:
        os._exit(0)  # (not for the single-GPU line: profilers such as rocprofv3 flush their traces at normal exit)


if __name__ == "__main__":
    main()
