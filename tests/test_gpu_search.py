"""GPU: the REAL PhyML tree search running on the device engine through the drop-in boundary.

oracle/_ref/phyml_glue_driver (built in the build container from the unmodified reference + oracle/glue_driver.c, see
its header) is PhyML's own SPR search and branch-length optimisation -- spr.c, optimiz.c, the traversals of lk.c --
with Lk / dLk / Update_Partial_Lk / Update_PMat_At_Given_Edge / Update_Eigen_Lr served by libphyhip.so through the C ABI.

  check mode   the reference's own arithmetic runs alongside and steers the search; EVERY scalar of the search (tens of
               thousands of Lk(b), ~100 000 dLk) is compared call by call with what the device returns: the worst
               relative difference must stay below 1e-10 (north star: 1e-6)
  device mode  the search is driven only by device results and must end in the neighbourhood of the reference's end
               point on the same machine (sanity check: the heuristic is chaotic in the last bits of every lnL)

The CPU-only end point recorded in the build container (tests/golden/search_expected.json) is a sanity anchor only: the
search trajectory depends on the last bits of libm's exp/log, which differ between host CPUs.
"""
import json
import os
import re
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

GLUE = os.path.join(ROOT, "oracle", "_ref", "phyml_glue_driver")
EXPECTED = json.load(open(os.path.join(GOLDEN, "search_expected.json")))
_cache = {}


def run_search(name, mode, tmp_path, device_pmat=False):
    key = (name, mode, device_pmat)
    if key in _cache:
        return _cache[key]
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built (needs the reference: make -C oracle ref in the build container)")
    e = EXPECTED[name]
    wd = os.path.join(str(tmp_path), mode + ("_dp" if device_pmat else ""))
    os.makedirs(wd, exist_ok=True)
    shutil.copy(os.path.join(GOLDEN, "examples_" + e["example"] + ".phy"), os.path.join(wd, e["example"]))
    env = dict(os.environ, GLUE_MODE=mode, GLUE_DEVICE_PMAT="1" if device_pmat else "0")
    r = subprocess.run([GLUE] + e["driver_opts"] + ["--", "-i", e["example"]] + e["phyml_args"], cwd=wd, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2000:]
    _cache[key] = json.loads(m.group(1))
    return _cache[key]


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_every_call_of_a_real_search_matches_the_reference(name, tmp_path):
    info = run_search(name, "check", tmp_path)
    assert info["calls"]["Lk"] > 10000 and info["calls"]["dLk"] > 10000 and info["calls"]["Update_Partial_Lk"] > 50000
    assert info["worst_rel_lnL"] < 1e-10, info
    assert info["worst_rel_dlnL"] < 1e-6, info
    assert info["lnL_final"] > info["lnL_init"] + 10.0
    # sanity anchor: the build container's CPU-only run of the same command ended in the same neighbourhood
    assert abs(info["lnL_final"] - EXPECTED[name]["lnL_final"]) < 2e-3 * abs(EXPECTED[name]["lnL_final"])


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_real_search_driven_by_the_device(name, tmp_path):
    ref = run_search(name, "check", tmp_path)       # what the reference does on this machine
    info = run_search(name, "device", tmp_path)     # the same search, device results only
    assert abs(info["lnL_init"] - ref["lnL_init"]) <= 1e-12 * abs(ref["lnL_init"])
    # The heuristic is chaotic in the last bits of every lnL (the device differs from the AVX path by ~1e-15 relative, as
    # two host CPUs differ from each other through libm): call counts, intermediate trees and even the local optimum
    # reached differ between runs -- the reference's own end point on examples/nucleic was -5580.1235 on one box and
    # -5583.0951 on another.  So this is the sanity check SURVEY 8d asks for, not the parity gate (that is check mode):
    # the device-driven search must end in the same neighbourhood (1e-3 relative; typically 1e-7).
    assert abs(info["lnL_final"] - ref["lnL_final"]) <= 1e-3 * abs(ref["lnL_final"]), (info["lnL_final"], ref["lnL_final"])
    assert info["lnL_final"] > info["lnL_init"] + 10.0
    for k in ("Lk", "dLk", "Update_Partial_Lk"):
        assert 0.5 * ref["calls"][k] < info["calls"][k] < 2.0 * ref["calls"][k]


def test_real_search_with_device_built_matrices(tmp_path):
    """Same, with the P-matrices built on the device from the eigen system (src/lk.c:2344 route)."""
    ref = run_search("search_nucleic_spr", "check", tmp_path)
    info = run_search("search_nucleic_spr", "device", tmp_path, device_pmat=True)
    assert abs(info["lnL_final"] - ref["lnL_final"]) <= 1e-3 * abs(ref["lnL_final"]), (info["lnL_final"], ref["lnL_final"])
    assert info["lnL_final"] > info["lnL_init"] + 10.0


def test_lg4x_mixture_analysis_check_mode(tmp_path):
    """The reference's LG4X mixture analysis (examples/lg4x: four class trees, free rates, SPR + parameter optimisation,
    XML mode) with the class trees mirrored on the device: every MIXT_Lk evaluation at the P-matrix level is repeated by
    phyhip_calculate_mixture_log_likelihood, every MIXT_dLk by phyhip_calculate_mixture_eigen_lnl_dlnl, over the four class
    instances, and compared (first 6000 evaluations)."""
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built (needs the reference: make -C oracle ref in the build container)")
    base = str(tmp_path)
    os.makedirs(os.path.join(base, "examples", "lg4x")); os.makedirs(os.path.join(base, "run"))
    for f in os.listdir(os.path.join(GOLDEN, "lg4x")):
        shutil.copy(os.path.join(GOLDEN, "lg4x", f), os.path.join(base, "examples", "lg4x", f))
    shutil.copy(os.path.join(GOLDEN, "examples_proteic.phy"), os.path.join(base, "examples", "proteic"))
    env = dict(os.environ, GLUE_MODE="check", GLUE_MAX_MIXT="6000")
    r = subprocess.run([GLUE, "--", "--xml=../examples/lg4x/lg4x_check.xml"], cwd=os.path.join(base, "run"), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2000:]
    info = json.loads(m.group(1))
    assert info["class_instances"] == 4
    assert info["calls"]["MIXT_Lk"] + info["calls"]["MIXT_dLk"] - info["calls"]["MIXT_skipped"] >= 6000
    assert info["calls"]["MIXT_Lk"] - info["calls"]["MIXT_skipped"] > 500 and info["calls"]["MIXT_dLk"] > 1000, info
    assert info["worst_rel_mixture_lnL"] < 1e-10, info
    assert info["worst_rel_mixture_dlnL"] < 1e-6, info


def test_lg4x_mixture_analysis_driven_by_the_device(tmp_path):
    """The same analysis with MIXT_Lk / MIXT_dLk SERVED by the device (class partials never computed on the host): the
    first 8000 evaluations of the run; the optimiser must be climbing from the initial -12496.58 like the reference does."""
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built (needs the reference: make -C oracle ref in the build container)")
    base = str(tmp_path)
    os.makedirs(os.path.join(base, "examples", "lg4x")); os.makedirs(os.path.join(base, "run"))
    for f in os.listdir(os.path.join(GOLDEN, "lg4x")):
        shutil.copy(os.path.join(GOLDEN, "lg4x", f), os.path.join(base, "examples", "lg4x", f))
    shutil.copy(os.path.join(GOLDEN, "examples_proteic.phy"), os.path.join(base, "examples", "proteic"))
    env = dict(os.environ, GLUE_MODE="device", GLUE_MAX_MIXT="8000")
    r = subprocess.run([GLUE, "--", "--xml=../examples/lg4x/lg4x_check.xml"], cwd=os.path.join(base, "run"), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2000:]
    info = json.loads(m.group(1))
    assert info["mode"] == "device" and info["class_instances"] == 4
    assert info["calls"]["MIXT_Lk"] + info["calls"]["MIXT_dLk"] >= 8000
    assert -12496.6 < info["best_full_lnL"] < -12300.0 and info["best_full_lnL"] > -12490.0, info
