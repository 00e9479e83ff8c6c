"""GPU: the REAL PhyML tree search running on the device engine through the drop-in boundary.

oracle/_ref/phyml_glue_driver (built in the build container from the unmodified reference + oracle/glue_driver.c, see
its header) is PhyML's own SPR search and branch-length optimisation -- spr.c, optimiz.c, the traversals of lk.c --
with Lk / dLk / Update_Partial_Lk / Update_PMat_At_Given_Edge / Update_Eigen_Lr served by libphyhip.so through the C ABI.

  check mode   the reference's own arithmetic runs alongside and steers the search; EVERY scalar of the search (tens of
               thousands of Lk(b), ~100 000 dLk) is compared call by call with what the device returns: the worst
               relative difference must stay below 1e-10 (north star: 1e-6)
  device mode  the search is driven only by device results and must follow the reference-steered run on the same machine
               (the driver gives PhyML a private rand() stream, see glue_driver.c: the runs of record make exactly the same
               calls and agree to 1e-13 at the end; a last-bit tie broken the other way would still be legitimate, hence
               the two-level assertion)

The CPU-only end point recorded in the build container (tests/golden/search_expected.json) is a sanity anchor only: the
search trajectory depends on the last bits of libm's exp/log, which differ between host CPUs.
"""
import json
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

GLUE = os.path.join(ROOT, "oracle", "_ref", "phyml_glue_driver")
EXPECTED = json.load(open(os.path.join(GOLDEN, "search_expected.json")))
_cache = {}


def run_search(name, mode, tmp_path, device_pmat=False, devices=None, extra=()):
    key = (name, mode, device_pmat, devices, tuple(extra))
    if key in _cache:
        return _cache[key]
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built (needs the reference: make -C oracle ref in the build container)")
    e = EXPECTED[name]
    wd = os.path.join(str(tmp_path), mode + ("_dp" if device_pmat else "") + ("_sh" if devices else "") + "".join(extra).replace("-", "_"))
    os.makedirs(wd, exist_ok=True)
    shutil.copy(os.path.join(GOLDEN, "examples_" + e["example"] + ".phy"), os.path.join(wd, e["example"]))
    env = dict(os.environ, GLUE_MODE=mode, GLUE_DEVICE_PMAT="1" if device_pmat else "0")
    if devices:
        env["GLUE_DEVICES"] = devices
    r = subprocess.run([GLUE] + e["driver_opts"] + ["--", "-i", e["example"]] + e["phyml_args"] + list(extra), cwd=wd, env=env,
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
    # The device differs from the AVX path by ~1e-15 relative per call; a decision of the heuristic that hangs on such a
    # difference may legitimately go the other way (two host CPUs differ from each other the same way through libm: the
    # reference's own end point on examples/nucleic was -5580.1235 on one box and -5583.0951 on another).  So: the end point
    # must be in the same neighbourhood (SURVEY 8d's sanity check), and when the trajectory is the same -- it was, call for
    # call, in every run since PhyML got its private rand() stream -- the end points must agree to 1e-9.
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump({"check": {k: ref[k] for k in ("lnL_final", "calls", "seconds")}, "device": {k: info[k] for k in ("lnL_final", "calls", "seconds")}},
                  open(os.path.join(out, "search_end_points_%s.json" % name), "w"))
    assert abs(info["lnL_final"] - ref["lnL_final"]) <= 1e-3 * abs(ref["lnL_final"]), (info["lnL_final"], ref["lnL_final"])
    assert info["lnL_final"] > info["lnL_init"] + 10.0
    for k in ("Lk", "dLk", "Update_Partial_Lk"):
        assert 0.5 * ref["calls"][k] < info["calls"][k] < 2.0 * ref["calls"][k]
    if info["calls"] == ref["calls"]:
        assert abs(info["lnL_final"] - ref["lnL_final"]) <= 1e-9 * abs(ref["lnL_final"])


def test_real_search_with_alias_subpatt(tmp_path):
    """`phyml --alias_subpatt` (src/cl.c:502 -> src/lk.c:1294-1296): the application's Alias_One_Subpatt runs
    where the reference calls it -- tens of thousands of times in a search -- and, as in the reference (CPU counterpart:
    tests/test_oracle_golden.py), not one number of the path depends on it: the device-driven search is the same search."""
    base = run_search("search_nucleic_spr", "device", tmp_path)
    info = run_search("search_nucleic_spr", "device", tmp_path, extra=("--alias_subpatt",))
    # (the reference switches tree->update_alias_subpatt on for parts of a search only: src/spr.c:1235-1352, src/utilities.c:1742-1818)
    assert info["alias_one_subpatt_calls_made_here"] > 10000 and info["calls"]["Update_Partial_Lk"] > 100000
    assert base["alias_one_subpatt_calls_made_here"] == 0
    assert info["calls"] == base["calls"]
    assert info["lnL_init"] == base["lnL_init"] and info["lnL_final"] == base["lnL_final"] and info["tree"] == base["tree"]
    chk = run_search("search_nucleic_spr", "check", tmp_path, extra=("--alias_subpatt",))
    assert chk["worst_rel_lnL"] < 1e-10 and chk["worst_rel_dlnL"] < 1e-6


def test_real_search_on_a_sharded_instance(tmp_path):
    """PhyML's real spr.c / optimiz.c on the MULTI-GPU form of the C ABI (GLUE_DEVICES: three pattern shards, one RCCL
    all-reduce behind every Lk / dLk): every scalar against the reference's own arithmetic in check mode, and the
    device-driven search must make the single-instance run's calls and end on its end point (the shard sums are added in
    another order, 1e-15 relative per call, so the same tolerance rules as above apply)."""
    import torch
    devs = "0,1,0" if torch.cuda.device_count() >= 2 else "0,0,0"
    chk = run_search("search_nucleic_spr", "check", tmp_path, devices=devs)
    assert chk["calls"]["Lk"] > 10000 and chk["worst_rel_lnL"] < 1e-10 and chk["worst_rel_dlnL"] < 1e-6, chk
    one = run_search("search_nucleic_spr", "device", tmp_path)
    info = run_search("search_nucleic_spr", "device", tmp_path, devices=devs)
    assert abs(info["lnL_init"] - one["lnL_init"]) <= 1e-12 * abs(one["lnL_init"])
    assert abs(info["lnL_final"] - one["lnL_final"]) <= 1e-3 * abs(one["lnL_final"])
    if info["calls"] == one["calls"]:
        assert abs(info["lnL_final"] - one["lnL_final"]) <= 1e-9 * abs(one["lnL_final"])


@pytest.mark.parametrize("name", ["search_nucleic_spr", "search_proteic_spr"])
def test_real_search_with_device_built_matrices(name, tmp_path):
    """Same, with the P-matrices built on the device from the eigen system (src/lk.c:2344 route).  Since round 6 the device's exp()
    is the reference's libm's (phyml_amd/csrc/phyhip_exp.hpp) and device-built matrices are the reference's doubles: the search
    driven through this route -- every SPR candidate a resident command, 20 states included, no upload -- makes exactly the calls of
    the run with uploaded host matrices and ends on the same double (and the calls of the run on the reference's own arithmetic)."""
    ref = run_search(name, "check", tmp_path)
    up = run_search(name, "device", tmp_path)                        # device-driven, host PMat() + upload
    info = run_search(name, "device", tmp_path, device_pmat=True)    # device-driven, matrices built on the device
    assert info["calls"] == up["calls"] and info["lnL_final"] == up["lnL_final"], (info, up)  # the same matrices: the same run
    assert info["calls"] == ref["calls"], (info["calls"], ref["calls"])
    # (against the CPU: the device adds the site terms in another order)
    assert abs(info["lnL_final"] - ref["lnL_final"]) <= 1e-11 * abs(ref["lnL_final"]), (info["lnL_final"], ref["lnL_final"])
    assert info["lnL_final"] > info["lnL_init"] + 10.0


@pytest.mark.parametrize("class_axis", [False, True], ids=["instance_per_class", "class_axis"])
def test_lg4x_mixture_analysis_check_mode(class_axis, tmp_path):
    """(class_axis: the four class trees on the category axis of ONE instance -- PHYHIP_FLAG_CLASS_AXIS, GLUE_CLASS_AXIS=1 --
    one traversal launch per mixture evaluation instead of four.)
    The reference's LG4X mixture analysis (examples/lg4x: four class trees, free rates, SPR + parameter optimisation,
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
    env = dict(os.environ, GLUE_MODE="check", GLUE_MAX_MIXT="6000", GLUE_CLASS_AXIS="1" if class_axis else "0")
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


@pytest.mark.parametrize("class_axis", [False, True], ids=["instance_per_class", "class_axis"])
def test_nucleotide_mixture_analysis_check_mode(class_axis, tmp_path):
    """The same for a four-class NUCLEOTIDE mixture (tests/golden/ntmix/nt4_check.xml: HKY85 + empirical frequencies / K80 +
    equal frequencies, four free rates, on examples/nucleic): the reference's own XML analysis, every MIXT_Lk / MIXT_dLk
    repeated on the device -- four one-category instances, or the four classes on the lanes of ONE class-axis instance."""
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built (needs the reference: make -C oracle ref in the build container)")
    base = str(tmp_path)
    os.makedirs(os.path.join(base, "examples", "ntmix")); os.makedirs(os.path.join(base, "run"))
    shutil.copy(os.path.join(GOLDEN, "ntmix", "nt4_check.xml"), os.path.join(base, "examples", "ntmix", "nt4_check.xml"))
    shutil.copy(os.path.join(GOLDEN, "examples_nucleic.phy"), os.path.join(base, "examples", "nucleic"))
    env = dict(os.environ, GLUE_MODE="check", GLUE_MAX_MIXT="4000", GLUE_CLASS_AXIS="1" if class_axis else "0")
    r = subprocess.run([GLUE, "--", "--xml=../examples/ntmix/nt4_check.xml"], cwd=os.path.join(base, "run"), env=env,
                       stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2000:]
    info = json.loads(m.group(1))
    assert info["class_instances"] == 4
    assert info["calls"]["MIXT_Lk"] + info["calls"]["MIXT_dLk"] - info["calls"]["MIXT_skipped"] >= 1000, info
    assert info["worst_rel_mixture_lnL"] < 1e-10, info
    assert info["worst_rel_mixture_dlnL"] < 1e-6, info


@pytest.mark.parametrize("class_axis", [False, True], ids=["instance_per_class", "class_axis"])
def test_leave_one_out_cross_validation_check_mode(class_axis, tmp_path):
    """`cv.type="maxfold"` (MIXT_Maxfold_Cv, src/mixt.c:4198-4290 -> src/cv.c:294-420): every (site, taxon) position is hidden
    (Init_Partial_Lk_Tips_Double_One_Character -> phyhip_set_tip_partials_at_pattern), the pendant edge optimised (MIXT_dLk /
    MIXT_Lk, compared call by call as in every check-mode run), the character restored, and the reader's input -- p_lk_left of the
    pendant edge at the hidden site, per class tree -- downloaded from the device and compared bit for bit (SURVEY 8f rank 3's
    last reader)."""
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built (needs the reference: make -C oracle ref in the build container)")
    base = str(tmp_path)
    os.makedirs(os.path.join(base, "examples", "ntmix")); os.makedirs(os.path.join(base, "run"))
    shutil.copy(os.path.join(GOLDEN, "ntmix", "nt4_cv.xml"), os.path.join(base, "examples", "ntmix", "nt4_cv.xml"))
    shutil.copy(os.path.join(GOLDEN, "examples_nucleic.phy"), os.path.join(base, "examples", "nucleic"))
    env = dict(os.environ, GLUE_MODE="check", GLUE_MAX_MIXT="4000", GLUE_CLASS_AXIS="1" if class_axis else "0")
    r = subprocess.run([GLUE, "--", "--xml=../examples/ntmix/nt4_cv.xml"], cwd=os.path.join(base, "run"), env=env,
                       stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2000:]
    info = json.loads(m.group(1))
    assert info["class_instances"] == 4
    assert info["tip_characters_rewritten"] > 200, info           # two rewrites per hidden position (hide, restore)
    assert info["cv_vectors_compared"] > 400 and info["cv_vector_mismatches"] == 0, info
    assert info["calls"]["MIXT_dLk"] > 1000, info
    assert info["worst_rel_mixture_lnL"] < 1e-10, info
    assert info["worst_rel_mixture_dlnL"] < 1e-6, info


@pytest.mark.parametrize("class_axis,device_pmat", [(False, False), (True, False), (True, True)],
                         ids=["instance_per_class", "class_axis", "class_axis_device_matrices"])
def test_lg4x_mixture_analysis_driven_by_the_device(class_axis, device_pmat, tmp_path):
    """The same analysis with MIXT_Lk / MIXT_dLk SERVED by the device (class partials never computed on the host): the
    first 8000 evaluations of the run; the optimiser must be climbing from the initial -12496.58 like the reference does."""
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built (needs the reference: make -C oracle ref in the build container)")
    base = str(tmp_path)
    os.makedirs(os.path.join(base, "examples", "lg4x")); os.makedirs(os.path.join(base, "run"))
    for f in os.listdir(os.path.join(GOLDEN, "lg4x")):
        shutil.copy(os.path.join(GOLDEN, "lg4x", f), os.path.join(base, "examples", "lg4x", f))
    shutil.copy(os.path.join(GOLDEN, "examples_proteic.phy"), os.path.join(base, "examples", "proteic"))
    env = dict(os.environ, GLUE_MODE="device", GLUE_MAX_MIXT="8000", GLUE_CLASS_AXIS="1" if class_axis else "0",
               GLUE_DEVICE_PMAT="1" if device_pmat else "0")
    r = subprocess.run([GLUE, "--", "--xml=../examples/lg4x/lg4x_check.xml"], cwd=os.path.join(base, "run"), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2000:]
    info = json.loads(m.group(1))
    assert info["mode"] == "device" and info["class_instances"] == 4
    assert info["calls"]["MIXT_Lk"] + info["calls"]["MIXT_dLk"] >= 8000
    assert -12496.6 < info["best_full_lnL"] < -12300.0 and info["best_full_lnL"] > -12490.0, info


def _read_interleaved(path):
    """PHYLIP interleaved alignment (the reference's examples/proteic as shipped) -> (names, sequences)."""
    lines = [l.rstrip("\n") for l in open(path)]
    n, L = (int(x) for x in lines[0].split())
    names, seqs, row = [], [], 0
    for l in lines[1:]:
        if not l.strip():
            continue
        if len(names) < n:
            names.append(l.split()[0]); seqs.append("".join(l.split()[1:]))
        else:
            seqs[row % n] += "".join(l.split()); row += 1
    assert all(len(s) == L for s in seqs), [len(s) for s in seqs][:3]
    return names, seqs


@pytest.mark.parametrize("mode,class_axis", [("check", False), ("check", True), ("device", True)],
                         ids=["check_instances", "check_class_axis", "device_class_axis"])
def test_partitioned_mixture_analysis(mode, class_axis, tmp_path):
    """A data partition with TWO elements (the two halves of the example alignment under a four-class and a two-class
    mixture, shared topology and branch lengths; tests/golden/lg4x/part2_check.xml): MIXT_Lk / MIXT_dLk are sums over the
    elements (src/mixt.c:862, 1161-1176, 3325-3360), every element being one instance group on the device.  Check mode:
    every evaluation of the reference's analysis against the device; device mode: the device serves them."""
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built (needs the reference: make -C oracle ref in the build container)")
    base = str(tmp_path)
    os.makedirs(os.path.join(base, "examples", "lg4x")); os.makedirs(os.path.join(base, "run"))
    for f in os.listdir(os.path.join(GOLDEN, "lg4x")):
        shutil.copy(os.path.join(GOLDEN, "lg4x", f), os.path.join(base, "examples", "lg4x", f))
    names, seqs = _read_interleaved(os.path.join(GOLDEN, "examples_proteic.phy"))
    for tag, (lo, hi) in (("part_a", (0, 270)), ("part_b", (270, 547))):
        with open(os.path.join(base, "examples", tag), "w") as f:
            f.write(f"{len(names)} {hi - lo}\n")
            for nm, sq in zip(names, seqs):
                f.write(f"{nm}  {sq[lo:hi]}\n")
    env = dict(os.environ, GLUE_MODE=mode, GLUE_MAX_MIXT="5000", GLUE_CLASS_AXIS="1" if class_axis else "0", GLUE_DEVICE_PMAT="0")
    r = subprocess.run([GLUE, "--", "--xml=../examples/lg4x/part2_check.xml"], cwd=os.path.join(base, "run"), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2500:]
    info = json.loads(m.group(1))
    assert info["class_instances"] == 6          # four + two class trees
    assert info["calls"]["MIXT_Lk"] + info["calls"]["MIXT_dLk"] >= 5000 and info["calls"]["MIXT_skipped"] == 0, info
    if mode == "check":
        assert info["worst_rel_mixture_lnL"] < 1e-10 and info["worst_rel_mixture_dlnL"] < 1e-6, info
    else:
        assert info["best_full_lnL"] > -13000.0 and np.isfinite(info["last_mixture_lnL"]), info


@pytest.mark.parametrize("mode,class_axis", [("check", False), ("check", True), ("device", True)],
                         ids=["check_instances", "check_class_axis", "device_class_axis"])
def test_invariant_mixture_analysis(mode, class_axis, tmp_path):
    """A +I mixture (tests/golden/lg4x/inv_check.xml: three classes + the invariant class): the invariant class has no
    class tree on the device; its share enters the combination (phyhip_set_mixture_invariant_sites, src/mixt.c:1079-1112,
    3212-3275).  Check mode compares every MIXT_Lk / MIXT_dLk of the reference's analysis, device mode serves them."""
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built (needs the reference: make -C oracle ref in the build container)")
    base = str(tmp_path)
    os.makedirs(os.path.join(base, "examples", "lg4x")); os.makedirs(os.path.join(base, "run"))
    for f in os.listdir(os.path.join(GOLDEN, "lg4x")):
        shutil.copy(os.path.join(GOLDEN, "lg4x", f), os.path.join(base, "examples", "lg4x", f))
    shutil.copy(os.path.join(GOLDEN, "examples_proteic.phy"), os.path.join(base, "examples", "proteic"))
    env = dict(os.environ, GLUE_MODE=mode, GLUE_MAX_MIXT="5000", GLUE_CLASS_AXIS="1" if class_axis else "0", GLUE_DEVICE_PMAT="0")
    r = subprocess.run([GLUE, "--", "--xml=../examples/lg4x/inv_check.xml"], cwd=os.path.join(base, "run"), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2500:]
    info = json.loads(m.group(1))
    assert 3 <= info["class_instances"] <= 4     # the computing class trees of the chain
    assert info["calls"]["MIXT_Lk"] + info["calls"]["MIXT_dLk"] >= 5000 and info["calls"]["MIXT_skipped"] == 0, info
    if mode == "check":
        assert info["worst_rel_mixture_lnL"] < 1e-10 and info["worst_rel_mixture_dlnL"] < 1e-6, info
    else:
        assert info["best_full_lnL"] > -13000.0 and np.isfinite(info["last_mixture_lnL"]), info


# ---- fast branch supports (src/alrt.c) on device-resident state: the download hooks of SURVEY 8(f) rank 3 -------------

SUPPORT_ARGS = ["-d", "nt", "-m", "GTR", "-f", "0.3,0.2,0.2,0.3", "-c", "4", "-a", "0.8", "-o", "n", "-b", "-4", "--r_seed", "1"]


def run_supports(mode, tmp_path):
    key = ("supports", mode)
    if key in _cache:
        return _cache[key]
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built")
    wd = os.path.join(str(tmp_path), "sup_" + mode)
    os.makedirs(wd, exist_ok=True)
    shutil.copy(os.path.join(GOLDEN, "examples_nucleic.phy"), os.path.join(wd, "nucleic"))
    r = subprocess.run([GLUE, "--gtr-rr", "1,2.5,0.8,1.2,3.0,1", "--", "-i", "nucleic"] + SUPPORT_ARGS, cwd=wd,
                       env=dict(os.environ, GLUE_MODE=mode), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2000:]
    _cache[key] = json.loads(m.group(1))
    return _cache[key]


def _numbers(nwk):
    """(support values, branch lengths, topology with the numbers removed) of a Newick string written by Write_Tree."""
    sup = [float(x) for x in re.findall(r"\)([0-9.eE+-]+):", nwk)]
    bl = [float(x) for x in re.findall(r":([0-9.eE+-]+)", nwk)]
    return sup, bl, re.sub(r"\)[0-9.eE+-]+:", "):", re.sub(r":[0-9.eE+-]+", ":", nwk))


def test_branch_supports_per_site_outputs_match_the_reference(tmp_path):
    """SH-like supports (`-b -4`): alrt.c reads c_lnL_sorted of the last Lk() for each of the three NNI configurations
    (src/alrt.c:453,555,682).  Check mode: after every device evaluation of the support phase the per-pattern outputs
    (log-likelihood, per-category likelihoods, scale exponent) are downloaded and compared with what Lk_Core just wrote."""
    info = run_supports("check", tmp_path)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump({k: v for k, v in info.items() if k != "tree"}, open(os.path.join(out, "supports_check_mode.json"), "w"))
    assert info["site_output_downloads"] > 1000, info
    assert info["worst_rel_site_output"] < 1e-9, info
    assert info["worst_rel_lnL"] < 1e-10 and info["worst_rel_dlnL"] < 1e-6, info
    assert info["support_tree"].count(")") > 40


def test_branch_supports_from_device_resident_state(tmp_path):
    """Device mode: the host buffers alrt.c reads are filled by phyhip_get_site_outputs only; supports, branch lengths and
    topology must come out as in the CPU-only run of the same command on this machine.  (The driver gives the reference a
    private rand() stream -- see glue_driver.c -- so the RELL resampling of Statistics_To_SH, src/alrt.c:1148, draws the
    same replicates in every mode; what is left are the ~1e-15 relative differences of the optimised NNI configurations,
    which can move a replicate or two out of 10 000 where configurations are almost tied.)"""
    host = run_supports("host", tmp_path)
    dev = run_supports("device", tmp_path)
    assert dev["site_output_downloads"] > 1000
    hs, hb, ht = _numbers(host["support_tree"])
    ds, db, dt = _numbers(dev["support_tree"])
    assert ht == dt and len(hs) == len(ds) > 40
    diffs = sorted(abs(a - b) for a, b in zip(hs, ds))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump({"host": hs, "device": ds, "host_seconds": host["seconds"], "device_seconds": dev["seconds"]},
                  open(os.path.join(out, "supports_host_vs_device.json"), "w"))
    assert diffs[-1] <= 0.005 and diffs[len(diffs) // 2] <= 2e-4, list(zip(hs, ds))
    assert max(abs(a - b) for a, b in zip(hb, db)) <= 1e-6
    assert abs(dev["lnL_final"] - host["lnL_final"]) <= 1e-12 * abs(host["lnL_final"])


# ---- ancestral reconstruction (src/ancestral.c) from downloaded partial vectors ---------------------------------------

def run_ancestral(mode, tmp_path):
    key = ("ancestral", mode)
    if key in _cache:
        return _cache[key]
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built")
    wd = os.path.join(str(tmp_path), "anc_" + mode)
    os.makedirs(wd, exist_ok=True)
    shutil.copy(os.path.join(GOLDEN, "examples_nucleic.phy"), os.path.join(wd, "nucleic"))
    args = [a for a in SUPPORT_ARGS]
    args[args.index("-b") + 1] = "0"
    r = subprocess.run([GLUE, "--gtr-rr", "1,2.5,0.8,1.2,3.0,1", "--", "-i", "nucleic"] + args + ["--ancestral", "--print_site_lnl"], cwd=wd,
                       env=dict(os.environ, GLUE_MODE=mode), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2000:]
    rows = []
    for line in open(os.path.join(wd, "nucleic_phyml_ancestral_seq.txt")):
        f = line.split()
        if len(f) == 7 and f[0].isdigit():
            rows.append((int(f[0]), int(f[1]), [float(x) for x in f[2:6]], f[6]))
    sites = []
    for line in open(os.path.join(wd, "nucleic_phyml_lk.txt")):
        f = line.split()
        if len(f) == 10 and f[0].isdigit():
            sites.append([float(x) for x in f])
    _cache[key] = (json.loads(m.group(1)), rows, sites)
    return _cache[key]


def test_ancestral_reconstruction_and_site_likelihoods_from_downloads(tmp_path):
    """`--ancestral --print_site_lnl`: ancestral.c walks the partial and scale vectors of every edge on the host (src/ancestral.c:677-869).
    Check mode: every device buffer is downloaded (phyhip_get_partials / phyhip_get_scale_factors) and compared BIT FOR BIT
    with the host vector the reference computed for the same edge side.  Device mode: the host vectors are filled from the
    downloads only and the reference's reconstruction must print the CPU-only run's marginal probabilities and states."""
    info = run_ancestral("check", tmp_path)[0]
    assert info["mirrored_buffers"] >= 2 * (2 * 54 - 3) - 54 and info["mirror_mismatches"] == 0, info
    (_, host, hsites), (dinfo, dev, dsites) = run_ancestral("host", tmp_path), run_ancestral("device", tmp_path)
    assert dinfo["mirrored_buffers"] == info["mirrored_buffers"]
    assert len(host) == len(dev) > 10000
    for h, d in zip(host, dev):
        assert h[0] == d[0] and h[1] == d[1] and h[3] == d[3], (h, d)
        for a, b in zip(h[2], d[2]):
            assert abs(a - b) <= 1e-5 * max(abs(a), 1e-30) + 1e-300, (h, d)
    # `--print_site_lnl` (src/io.c:1870-2013) of the same runs: site likelihood, scale exponent, per-category likelihoods,
    # posterior mean rate per alignment column, from phyhip_get_site_outputs
    assert len(hsites) == len(dsites) > 800
    for h, d in zip(hsites, dsites):
        for a, b in zip(h, d):
            assert abs(a - b) <= 1e-5 * abs(a), (h, d)


# ---- bootstrap (src/utilities.c:3884-4110): a new tree object per replicate on resampled pattern weights -----------------

def run_bootstrap(mode, tmp_path):
    key = ("bootstrap", mode)
    if key in _cache:
        return _cache[key]
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/phyml_glue_driver not built")
    wd = os.path.join(str(tmp_path), "boot_" + mode)
    os.makedirs(wd, exist_ok=True)
    shutil.copy(os.path.join(GOLDEN, "examples_nucleic.phy"), os.path.join(wd, "nucleic"))
    args = ["-d", "nt", "-m", "GTR", "-f", "0.3,0.2,0.2,0.3", "-c", "4", "-a", "0.8", "-o", "l", "-b", "4", "--r_seed", "1"]
    r = subprocess.run([GLUE, "--gtr-rr", "1,2.5,0.8,1.2,3.0,1", "--", "-i", "nucleic"] + args, cwd=wd,
                       env=dict(os.environ, GLUE_MODE=mode), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2000:]
    _cache[key] = json.loads(m.group(1))
    return _cache[key]


def test_bootstrap_replicates_on_resampled_weights(tmp_path):
    """`-b 4 -o l`: every replicate is a fresh tree object that aliases the original's buffers, rewrites the tips and
    carries resampled weights; the driver answers with a fresh instance per replicate (weights and tips uploaded again).
    Check mode: every scalar of the four replicate optimisations against the reference's own; device mode: the CPU-only
    run's bootstrap counts, topology and branch lengths."""
    chk = run_bootstrap("check", tmp_path)
    assert chk["instances_created"] >= 6, chk
    assert chk["worst_rel_lnL"] < 1e-10 and chk["worst_rel_dlnL"] < 1e-6, chk
    host, dev = run_bootstrap("host", tmp_path), run_bootstrap("device", tmp_path)
    assert dev["calls"] == host["calls"]
    hs, hb, ht = _numbers(host["support_tree"])
    ds, db, dt = _numbers(dev["support_tree"])
    assert ht == dt and hs == ds and len(hs) > 40 and max(hs) == 4.0
    assert max(abs(a - b) for a, b in zip(hb, db)) <= 1e-7
    assert abs(dev["lnL_final"] - host["lnL_final"]) <= 1e-10 * abs(host["lnL_final"])
