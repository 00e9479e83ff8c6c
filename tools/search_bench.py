"""End-to-end datapoint: PhyML's real SPR search (oracle/_ref/phyml_glue_driver, see oracle/glue_driver.c) on a synthetic
alignment, driven by the device engine vs CPU-only (the reference's own AVX path, 1 core) on the same box.
usage: python tools/search_bench.py [n_taxa] [n_patterns] [--skip-host]"""
import json, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phyml_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
skip_host = "--skip-host" in sys.argv
aa = "--aa" in sys.argv
ns = 20 if aa else 4
GLUE = os.path.join(ROOT, "oracle", "_ref", "phyml_glue_driver")
tmp = tempfile.mkdtemp(prefix="search_")
tree = synth.random_tree(n, 11, 0.02, 0.15)
st = synth.simulate_states(tree, P, ns, 11)
synth.write_phylip(os.path.join(tmp, "ali.phy"), tree.names, synth.states_to_chars(st, ns))
if aa:
    args = ["--", "-i", "ali.phy", "-d", "aa", "-m", "LG", "-f", "m", "-c", "4", "-a", "0.8", "-s", "SPR", "-o", "tl", "-b", "0",
            "--r_seed", "1", "--no_colalias"]
else:
    args = ["--gtr-rr", "1,2.5,0.8,1.2,3.0,1", "--", "-i", "ali.phy", "-d", "nt", "-m", "GTR", "-f", "0.3,0.2,0.2,0.3", "-c", "4", "-a", "0.8",
            "-s", "SPR", "-o", "tl", "-b", "0", "--r_seed", "1", "--no_colalias"]
out = {"taxa": n, "patterns": P, "states": ns}
for mode in (["device"] if skip_host else ["device", "host"]):
    t0 = time.time()
    wrap = os.environ.get("SEARCH_BENCH_WRAP", "").split()  # e.g. "rocprofv3 --kernel-trace --stats -d /x --": profile the driver itself
    r = subprocess.run(wrap + [GLUE] + args, cwd=tmp, env=dict(os.environ, GLUE_MODE=mode), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    if not m:
        print(r.stdout[-1500:]); raise SystemExit(1)
    info = json.loads(m.group(1)); info.pop("tree")
    out[mode] = info
if "host" in out:
    out["speedup"] = out["host"]["seconds"] / out["device"]["seconds"]
print(json.dumps(out))
