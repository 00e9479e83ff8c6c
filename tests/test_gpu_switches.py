"""Every A/B switch of the engine (another kernel or another route to the same numbers; read by the DIAG build only --
phyml_amd/lib_diag, built next to the product by __graft_entry__.build()) gives the same numbers: golden lnL, partial vectors
and scale vectors bit-equal to the oracle, and a seeded SPR / Br_Len_Opt call stream equal to the default's, scalar by
scalar (tools/README.md lists the switches and what each was measured for).  The switches the PRODUCT library still reads
(PHYHIP_RESIDENT, PHYHIP_HOST_SUM) are run on the product."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = [{}, {"PHYHIP_NT_GROUPS": "1"}, {"PHYHIP_NT_GROUPS": "2"}, {"PHYHIP_NT_GROUPS": "4"}, {"PHYHIP_NT2_DIST": "1"},
        {"PHYHIP_NT2_DIST": "1", "PHYHIP_NT_GROUPS": "1"}, {"PHYHIP_FOLD_PMATS": "0"}, {"PHYHIP_ARGS_RECS": "0"},
        {"PHYHIP_SPIN": "0"}, {"PHYHIP_EAGER_PMAT": "0"}, {"PHYHIP_PM_COPY": "1"}, {"PHYHIP_GENERIC_NT": "1"},
        {"PHYHIP_GENERIC_AA": "1"}, {"PHYHIP_AA_NW": "3"}, {"PHYHIP_SPLIT_REDUCE": "1"}, {"PHYHIP_ARG_UPLOADS": "0"},
        {"PHYHIP_FUSE_EIGEN": "0"}, {"PHYHIP_FOLD_GRID": "8"}, {"PHYHIP_DLK_GRID": "7"}, {"PHYHIP_BIG_DEVICE_SUM": "100000000"},
        {"PHYHIP_BIG_GROUP_SUM": "0"}, {"PHYHIP_BIG_ONE_SHOT": "0"}, {"PHYHIP_PUSH_CMDS": "0"}, {"PHYHIP_PUSH_CMDS": "2"},
        {"PHYHIP_PUSH_NO_MOVDIR": "1"},
        {"PHYHIP_PMAT_THREADS": "1024"}, {"PHYHIP_RESIDENT_DIRECT": "4"}, {"PHYHIP_VIRT_INLINE": "0"}, {"PHYHIP_VIRT_MIN_OPS": "0"}, {"PHYHIP_NT_MIXED": "0"},
        {"PHYHIP_PMAT20": "0"}, {"PHYHIP_PMAT20": "1"}, {"PHYHIP_AA_NT": "2"}, {"PHYHIP_AA_D2": "1"}]
PRODUCT = [{}, {"PHYHIP_RESIDENT": "0"}, {"PHYHIP_HOST_SUM": "0"}]
_cache = {}


def _ids(sw):
    return ",".join(f"{k[7:]}={v}" for k, v in sw.items()) or "default"


def _run(libdir, sw):
    key = (libdir, tuple(sorted(sw.items())))
    if key not in _cache:
        env = dict(os.environ)
        env.update(sw)
        env["PHYHIP_LIBDIR"] = os.path.join(ROOT, "phyml_amd", libdir)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_runner.py")], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [x for x in r.stdout.splitlines() if x.startswith("SWITCH_RESULT ")][-1]
        _cache[key] = json.loads(line[len("SWITCH_RESULT "):])
    return _cache[key]


def _check(libdir, sw):
    res, base = _run(libdir, sw), _run(libdir, {})
    assert res["lnl_rel"] < 1e-12 and res["vectors_bit_equal"] and res["aa_lnl_rel"] < 1e-12
    # device-built 20-state matrices: the same bits whichever kernel / block size builds them; the lnL on them as well unless the
    # switch selects another 20-state traversal kernel or another grouping of its block sums
    assert res["aa_device_matrices"]["sha256"] == base["aa_device_matrices"]["sha256"]
    if "PHYHIP_GENERIC_AA" not in sw and "PHYHIP_AA_NW" not in sw and "PHYHIP_SPLIT_REDUCE" not in sw and "PHYHIP_HOST_SUM" not in sw:
        assert res["aa_device_matrices"]["lnL"] == base["aa_device_matrices"]["lnL"], sw
    reorder = ("PHYHIP_GENERIC_NT" in sw or "PHYHIP_HOST_SUM" in sw or "PHYHIP_SPLIT_REDUCE" in sw or "PHYHIP_DLK_GRID" in sw or
               sw.get("PHYHIP_NT_GROUPS") in ("1", "4") or "PHYHIP_NT_MIXED" in sw or "PHYHIP_ARGS_RECS" in sw)
    # (PHYHIP_NT_MIXED=0: one wave shape -- the 9 000-pattern stream's full traversals sum other blocks; PHYHIP_ARGS_RECS=0: its
    # short launches go through the list form, i.e. the two-shape launch, instead of the one-shape argument form)
    # ... and for the 20-state stream: another traversal kernel, another number of wave-tiles per workgroup, another final sum
    reorder_aa = ("PHYHIP_GENERIC_AA" in sw or "PHYHIP_AA_NW" in sw or "PHYHIP_HOST_SUM" in sw or "PHYHIP_SPLIT_REDUCE" in sw or
                  "PHYHIP_DLK_GRID" in sw)
    hexl = lambda r, k: [float.fromhex(x) for x in r[k]]
    for k, ro in (("stream", reorder), ("stream_host", reorder), ("stream_big", reorder), ("stream_aa", reorder_aa)):
        u, v = hexl(res, k), hexl(base, k)
        ud, vd = hexl(res, k + "_d" if k != "stream" else "stream2"), hexl(base, k + "_d" if k != "stream" else "stream2")
        if ro:  # another kernel shape / another final sum adds the patterns' contributions in another order
            # log-likelihoods (|lnL| ~ 1e4): relative, the bar of the parity tests
            assert max(abs(x - y) / abs(y) for x, y in zip(u, v) if y != 0.0) < 1e-12, (k, sw)
            # derivatives: a sum of sum(w) per-pattern terms of either sign, each O(1) to O(100) -- reordering it moves the result
            # by rounding errors of the partial sums, not of the (possibly tiny) total: an absolute bar of 1e-13 per unit of weight
            assert max(abs(x - y) for x, y in zip(ud, vd)) <= 1e-13 * res["sum_w"][k], (k, sw)
        else:
            assert u == v and ud == vd, (k, sw)


@pytest.mark.parametrize("sw", DIAG, ids=_ids)
def test_diag_switch_gives_the_same_numbers(sw):
    _check("lib_diag", sw)


@pytest.mark.parametrize("sw", PRODUCT, ids=_ids)
def test_product_switch_gives_the_same_numbers(sw):
    _check("lib", sw)


def test_the_diag_build_and_the_product_agree():
    a, b = _run("lib_diag", {}), _run("lib", {})
    assert a["stream"] == b["stream"] and a["stream2"] == b["stream2"]
