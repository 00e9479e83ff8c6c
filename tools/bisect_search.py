#!/usr/bin/env python3
"""tools/bisect_search.py <name> [mode]: one of the real PhyML searches of tests/test_gpu_search.py (tests/golden/search_expected.json)
through the glue driver with the DIAG build preloaded, under several settings of the virtual-buffer switches -- which of them
changes the worst disagreement with the reference (check mode).  Developer aid."""
import json, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
GLUE = os.path.join(ROOT, "oracle", "_ref", "phyml_glue_driver")
name = sys.argv[1] if len(sys.argv) > 1 else "search_proteic_spr"
mode = sys.argv[2] if len(sys.argv) > 2 else "check"
e = json.load(open(os.path.join(GOLDEN, "search_expected.json")))[name]
for label, sw in (("default", {}), ("no in-step children", {"PHYHIP_VIRT_INLINE": "0"}), ("every buffer stored", {"PHYHIP_VIRT_MIN_OPS": "0"}),
                  ("threshold 1000", {"PHYHIP_VIRT_MIN_OPS": "1000"})):
    wd = tempfile.mkdtemp(prefix="bisect_")
    shutil.copy(os.path.join(GOLDEN, "examples_" + e["example"] + ".phy"), os.path.join(wd, e["example"]))
    env = dict(os.environ, GLUE_MODE=mode, GLUE_DEVICE_PMAT="0", LD_PRELOAD=os.path.join(ROOT, "phyml_amd", "lib_diag", "libphyhip.so"), **sw)
    r = subprocess.run([GLUE] + e["driver_opts"] + ["--", "-i", e["example"]] + e["phyml_args"], cwd=wd, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    if not m:
        print(label, "-> rc", r.returncode, r.stdout[-300:].replace("\n", " | "))
        continue
    info = json.loads(m.group(1))
    print(label, "-> worst_rel_lnL", info["worst_rel_lnL"], "worst_rel_dlnL", info["worst_rel_dlnL"], "lnL_final", info["lnL_final"], "seconds", info["seconds"])
