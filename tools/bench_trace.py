"""Wall time of replaying a recorded PhyML search prefix (tests/golden/trace_*.phyg) on the device engine."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from phyml_amd import phyg, replay
import test_gpu_trace as T

name = sys.argv[1] if len(sys.argv) > 1 else "trace_nucleic_spr"
host_pmat = (sys.argv[2] != "device") if len(sys.argv) > 2 else True
d = phyg.load(os.path.join(ROOT, "tests", "golden", name + ".phyg"))
tr, ro, ro2 = replay.recorded_trace(d)
t = T.device_tree_from_recorded(d, host_pmat)
t.Replay_Surface_Trace(tr)
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    out, out2 = t.Replay_Surface_Trace(tr)
dt = (time.perf_counter() - t0) / reps
k = tr["kind"]
nsc = int(np.isin(k, (2, 4, 5)).sum())
print(f"{name} host_pmat={host_pmat}: {len(k)} records, {nsc} scalar returns: {dt*1e3:.1f} ms = {dt/len(k)*1e6:.2f} us/record, {dt/nsc*1e6:.1f} us per scalar-returning call")
t.close()
