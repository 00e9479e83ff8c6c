#!/usr/bin/env python3
"""tools/host_share.py [n_taxa] [n_patterns] [--device-pmat]: PhyML's real SPR search through the glue driver (oracle/glue_driver.c) with
GLUE_HOSTPROF=1: how much of the run's wall time the shim's own host code takes -- per entry point of the C ABI (ns per call, share of
the run) and per interposed surface function around it -- against the device waits and the reference's own host work.  Prints the
GLUE_HOSTPROF line, the call counts and a markdown table (profiles/r05_host_share.md)."""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phyml_amd import synth

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(pos[0]) if pos else 150
P = int(pos[1]) if len(pos) > 1 else 20000
GLUE = os.path.join(ROOT, "oracle", "_ref", "phyml_glue_driver")
tmp = tempfile.mkdtemp(prefix="hostshare_")
tree = synth.random_tree(n, 11, 0.02, 0.15)
st = synth.simulate_states(tree, P, 4, 11)
synth.write_phylip(os.path.join(tmp, "ali.phy"), tree.names, synth.states_to_chars(st, 4))
args = ["--gtr-rr", "1,2.5,0.8,1.2,3.0,1", "--", "-i", "ali.phy", "-d", "nt", "-m", "GTR", "-f", "0.3,0.2,0.2,0.3", "-c", "4", "-a", "0.8",
        "-s", "SPR", "-o", "tl", "-b", "0", "--r_seed", "1", "--no_colalias"]
env = dict(os.environ, GLUE_MODE="device", GLUE_HOSTPROF="1")
if "--device-pmat" in sys.argv:
    env["GLUE_DEVICE_PMAT"] = "1"
r = subprocess.run([GLUE] + args, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
hp = json.loads(re.search(r"GLUE_HOSTPROF (\{.*\})", r.stdout).group(1))
info = json.loads(re.search(r"GLUE_DRIVER (\{.*\})", r.stdout).group(1)); info.pop("tree", None); info.pop("support_tree", None)
print(json.dumps({"taxa": n, "patterns": P, "host_profile": hp, "run": info}))
print(f"\n| {n} taxa x {P} nt patterns, device-driven SPR search: {hp['seconds']:.1f} s | calls | ns per call | seconds | share of the run |")
print("|---|---|---|---|---|")
for k, v in hp["slots"].items():
    print(f"| {k} | {v['calls']} | {v['ns_per_call']:.0f} | {v['seconds']:.3f} | {100 * v['share']:.2f} % |")
