#!/usr/bin/env python3
"""tools/search_check.py [n_taxa] [n_patterns] [--aa] [--device-pmat]: PhyML's real SPR search on a synthetic alignment in the glue driver's
CHECK mode (oracle/glue_driver.c): the reference computes everything itself and every Lk / dLk it returns is compared with the device's
value for the same call -- prints the worst relative differences and the call counts.  A larger sibling of tests/test_gpu_search.py."""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phyml_amd import synth
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(pos[0]) if pos else 80
P = int(pos[1]) if len(pos) > 1 else 4000
aa = "--aa" in sys.argv
ns = 20 if aa else 4
GLUE = os.path.join(ROOT, "oracle", "_ref", "phyml_glue_driver")
tmp = tempfile.mkdtemp(prefix="searchcheck_")
tree = synth.random_tree(n, 11, 0.02, 0.15)
st = synth.simulate_states(tree, P, ns, 11)
synth.write_phylip(os.path.join(tmp, "ali.phy"), tree.names, synth.states_to_chars(st, ns))
if aa:
    args = ["--", "-i", "ali.phy", "-d", "aa", "-m", "LG", "-f", "m", "-c", "4", "-a", "0.8", "-s", "SPR", "-o", "tl", "-b", "0", "--r_seed", "1", "--no_colalias"]
else:
    args = ["--gtr-rr", "1,2.5,0.8,1.2,3.0,1", "--", "-i", "ali.phy", "-d", "nt", "-m", "GTR", "-f", "0.3,0.2,0.2,0.3", "-c", "4", "-a", "0.8",
            "-s", "SPR", "-o", "tl", "-b", "0", "--r_seed", "1", "--no_colalias"]
env = dict(os.environ, GLUE_MODE="check", GLUE_FIRST_BAD="1", GLUE_DEVICE_PMAT="1" if "--device-pmat" in sys.argv else "0")
r = subprocess.run([GLUE] + args, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
if not m:
    print(r.stdout[-1500:]); raise SystemExit(1)
info = json.loads(m.group(1)); info.pop("tree", None); info.pop("support_tree", None)  # ("virtual_buffers": now | stores skipped | recomputed | stored on demand)
bad = [l for l in r.stdout.splitlines() if "GLUE_FIRST_BAD" in l]
print(json.dumps({"taxa": n, "patterns": P, "states": ns, "check": info, "first_bad": bad[:1]}))
