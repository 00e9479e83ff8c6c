#!/usr/bin/env python3
"""tools/step_breakdown.py: cfg2's Lk(NULL) step with and without per-launch HIP events (bench.py times with them on), and --
diag build, PHYHIP_HOSTPROF=1 -- the host's share per step (queue -> launch preparation, launch call, wait).  Developer tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from phyml_amd import workloads  # noqa: E402

wl = workloads.make("cfg2_nt_100x50k")
t = bench.build_tree(wl, device=0)
for _ in range(10):
    t.Lk(None)
out = {}
for prof in (0, 1, 0, 1):
    t.inst.profile(prof)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        t.Lk(None)
    torch.cuda.synchronize()
    out.setdefault(prof, []).append((time.perf_counter() - t0) / 200 * 1e6)
    if prof:
        ms, n, _ = t.inst.profile_read()
        out.setdefault("kernel", []).append(ms / n * 1e3)
t.inst.profile(0)
t.close()
print({k: [round(x, 2) for x in v] for k, v in out.items()})
