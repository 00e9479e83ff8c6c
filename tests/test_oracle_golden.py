"""Pin the CPU oracle (oracle/phylk_oracle.c) to golden vectors dumped from the REAL reference.

Known answers (reference AVX build):
    examples/nucleic GTR+G4 a=1  BioNJ  lnL = -5870.539924060607518
    examples/proteic LG+G4  a=1  BioNJ  lnL = -12569.851977923575760
Everything else in tests/golden/*.phyg was produced by oracle/ref_driver.c linked against the
reference's own objects (tests/golden/make_golden.py holds the commands).
"""
import os

import numpy as np
import pytest

import orc
from conftest import FIXTURES

KNOWN = {"nucleic_gtr_g4": -5870.539924060607518, "proteic_lg_g4": -12569.851977923575760}


@pytest.fixture(scope="module")
def evaluated(golden):
    cache = {}

    def get(name, arith=1):
        key = (name, arith)
        if key not in cache:
            d = golden(name)
            t = orc.tree_from_golden(d, arith=arith)
            lnl = t.lk(None, both_sides=True)
            cache[key] = (d, t, lnl, t.c_lnL_sorted.copy(), t.fact_sum_scale.copy(), t.unscaled_site_lk_cat.copy(),
                          t.cur_site_lk.copy())
        return cache[key]
    return get


@pytest.mark.parametrize("name", list(KNOWN))
def test_known_answer_lnl(name, evaluated):
    d, t, lnl, *_ = evaluated(name)
    assert abs(lnl - KNOWN[name]) / abs(KNOWN[name]) < 1e-14
    assert abs(d["lnL"][0] - KNOWN[name]) / abs(KNOWN[name]) < 1e-14


@pytest.mark.parametrize("name", FIXTURES)
def test_tip_encoding(name, golden):
    d = golden(name)
    m = orc.Model(d)
    for k in range(int(d["n_otu"][0])):
        v, ds, amb = orc.init_tip(m.datatype, d["tip_chars"][k])
        mask = (v * (1 << np.arange(m.ns))).sum(1).astype(np.int64)
        assert np.array_equal(mask, d["tip_mask"][k])
        assert np.array_equal(amb, d["tip_is_ambigu"][k])
        una = amb == 0
        assert np.array_equal(ds[una], d["tip_d_state"][k][una])


@pytest.mark.parametrize("name", FIXTURES)
def test_pmatrices_bit_exact(name, golden):
    d = golden(name)
    t = orc.tree_from_golden(d)
    npm = d["Pij_rr"].shape[0]
    assert npm > 0
    assert np.array_equal(t.pm[:npm], d["Pij_rr"])
    # rows are renormalised, entries floored at 1e-100 (src/models.c:293-298)
    assert np.all(t.pm >= 1e-100 * (1 - 1e-12))
    assert np.allclose(t.pm.sum(-1), 1.0, atol=1e-14)


@pytest.mark.parametrize("name", FIXTURES)
def test_lnl_and_site_outputs(name, evaluated):
    d, t, lnl, site, fact, unscaled, cur = evaluated(name)
    w = d["wght"] > 0
    assert abs(lnl - d["lnL"][0]) / abs(d["lnL"][0]) < 1e-14
    assert np.array_equal(fact[w], d["fact_sum_scale"][w])
    # every per-site output the reference keeps, bit for bit (round 6: the tip branch of one_class no longer lets the compiler fuse
    # a multiply-add the reference's binary does not have)
    assert np.array_equal(site[w], d["c_lnL_sorted"][w])
    assert np.array_equal(unscaled[w], d["unscaled_site_lk_cat"][w])
    assert np.array_equal(cur[w], d["cur_site_lk"][w])


@pytest.mark.parametrize("name", FIXTURES)
def test_partials_bit_exact_and_digests(name, evaluated):
    d, t, *_ = evaluated(name)
    w = d["wght"] > 0
    n_full = 0
    for e in d["full_edges"]:
        for side, nm in ((0, "left"), (1, "rght")):
            key = f"p_lk_{nm}_{e}"
            if key in d:
                assert np.array_equal(t.plk[(int(e), side)][w], d[key][w]), (name, e, nm)
                assert np.array_equal(t.scale[(int(e), side)][w], d[f"sum_scale_{nm}_{e}"][w])
                n_full += 1
    assert n_full >= 1
    # every edge side of the tree through its digest (sum, sum of squares, sum of scale exponents)
    n_chk = 0
    for e in range(t.ne):
        for side in (0, 1):
            if d["side_valid"][e, side]:
                a = t.plk[(e, side)][w]
                got = np.array([a.sum(), (a * a).sum(), t.scale[(e, side)][w].sum()])
                assert np.allclose(got, d["side_digest"][e, side], rtol=1e-11, atol=0), (name, e, side)
                n_chk += 1
    assert n_chk == 3 * (t.n - 2)   # every internal edge side: 2(2n-3) sides minus n tip sides


@pytest.mark.parametrize("name", FIXTURES)
def test_lnl_at_every_edge(name, evaluated):
    """Pulley principle with both sides filled: Lk(b) for every b (src/lk.c:2642-2684)."""
    d, t, *_ = evaluated(name)
    got = np.array([t.lk(e, refresh_pmat=False) for e in range(t.ne)])
    assert np.max(np.abs(got - d["edge_lnL"]) / np.abs(d["edge_lnL"])) < 1e-13


@pytest.mark.parametrize("name", FIXTURES)
def test_eigen_lr_and_dlk(name, evaluated):
    d, t, *_ = evaluated(name)
    w = d["wght"] > 0
    for k, e in enumerate(d["eigen_edges"]):
        e = int(e)
        t.lk(e, refresh_pmat=False)          # leaves fact_sum_scale of this edge behind
        t.update_eigen_lr(e)
        ref = d[f"dot_prod_{e}"]
        assert np.allclose(t.dot_prod[w], ref[w], rtol=1e-12, atol=1e-300)
        assert np.array_equal(t.fact_sum_scale[w], d[f"eig_fact_sum_scale_{e}"][w])
        for j in range(3):
            l_in, lnl_ref, dlnl_ref = d["dlk_triples"][k, j]
            l_out, lnl, dlnl = t.dlk(l_in)
            assert l_out == l_in
            assert abs(lnl - lnl_ref) / abs(lnl_ref) < 1e-13
            assert abs(dlnl - dlnl_ref) <= 1e-10 * max(1.0, abs(dlnl_ref))
        lnl_e = t.lk_eigen(t.len[e])
        assert abs(lnl_e - d[f"eig_lnL_{e}"][0]) / abs(lnl_e) < 1e-13


@pytest.mark.parametrize("name", ["nucleic_gtr_g4", "synth_aa_90x24"])
def test_scalar_order_agrees(name, evaluated):
    """The reference's scalar kernels (no FMA) and its AVX kernels agree to ~1e-15."""
    d, t, lnl, *_ = evaluated(name, arith=1)
    _, t0, lnl0, *_ = evaluated(name, arith=0)
    assert abs(lnl - lnl0) / abs(lnl) < 1e-13
    assert np.array_equal(t.fact_sum_scale, t0.fact_sum_scale)


def test_rescaling_is_exercised(golden):
    assert set(np.unique(golden("synth_nt_300x40")["fact_sum_scale"])) >= {0, 256, 512}
    assert 256 in np.unique(golden("synth_aa_90x24")["fact_sum_scale"])
    assert (golden("nucleic_zero_w")["wght"] == 0).sum() > 50
    assert golden("nucleic_gtr_g4_inv")["invar_model"][0] == 1


def test_update_order_counts(golden):
    d = golden("nucleic_gtr_g4")
    t = orc.tree_from_golden(d)
    t.lk(None, both_sides=False)
    assert t.n_updates == t.n - 2              # n-2 site-updates per pattern, SURVEY 8d
    t.n_updates = 0
    t.lk(None, both_sides=True)
    assert t.n_updates == 3 * (t.n - 2)


def test_generic_loop_door_is_pinned(golden):
    """`phyml --cov` (src/cl.c:753-757 -> mod->use_m4mod -> src/lk.c:1303-1324) is the one door through which the reference's
    `phyml` program reaches Update_Partial_Lk_Generic (src/lk.c:1332-1587) and returns a likelihood: 4-state data through the
    plain C loop.  tests/golden/nucleic_cov_generic.phyg (tests/golden/make_cov.py) is the same command as nucleic_gtr_g4_inv
    plus --cov.  The loop's arithmetic is the default path's (fused multiply-add chains) WITHOUT the all-ones shortcut of the
    SIMD kernels (src/avx.c:575-587): fully ambiguous subtrees yield rounded row sums instead of exactly 1.0 -- 51 of the 382
    patterns differ in the last bits.  The restatement with arith = 2 reproduces the dumped partial vector bit for bit; with
    the SIMD arithmetic it does not."""
    d, ref = golden("nucleic_cov_generic"), golden("nucleic_gtr_g4_inv")
    w = d["wght"] > 0
    assert np.array_equal(d["Pij_rr"], ref["Pij_rr"]) and not np.array_equal(d["p_lk_left_0"], ref["p_lk_left_0"])
    assert (np.asarray(d["p_lk_left_0"]) != np.asarray(ref["p_lk_left_0"])).any(axis=1).sum() == 51
    t = orc.tree_from_golden(d, arith=2)
    lnl = t.lk(None, both_sides=True)
    assert abs(lnl - d["lnL"][0]) / abs(lnl) < 1e-14
    assert np.array_equal(t.plk[(0, 0)][w], np.asarray(d["p_lk_left_0"])[w])
    assert np.array_equal(t.scale[(0, 0)][w], np.asarray(d["sum_scale_left_0"])[w])
    assert np.array_equal(t.fact_sum_scale, d["fact_sum_scale"])
    assert np.max(np.abs(t.c_lnL_sorted - d["c_lnL_sorted"]) / np.abs(d["c_lnL_sorted"])) < 1e-14
    for e in range(d["side_digest"].shape[0]):
        for side in range(2):
            if d["side_valid"][e, side]:
                p, sc = t.plk[(e, side)][w], t.scale[(e, side)][w]
                got = np.array([p.sum(), (p * p).sum(), float(sc.sum())])
                assert np.allclose(got, d["side_digest"][e, side], rtol=1e-12, atol=0), (e, side)
    t1 = orc.tree_from_golden(d, arith=1)
    t1.lk(None, both_sides=True)
    assert not np.array_equal(t1.plk[(0, 0)][w], np.asarray(d["p_lk_left_0"])[w])  # (the SIMD arithmetic is another one)


def test_alias_subpatt_changes_no_number_of_the_reference(tmp_path):
    """`phyml --alias_subpatt` (src/cl.c:502; SURVEY 8 row a2: Update_Partial_Lk calls Alias_One_Subpatt first, src/lk.c:1294-1296).
    The function fills the edges' patt_id / p_lk_loc arrays (src/utilities.c:13547-13666) and no likelihood function indexes them
    (`grep 'p_lk_loc[a-z_]*\\['` over src/: lk.c:2501-2502 and Alias_One_Subpatt itself, all writes).  Evidence from the reference
    itself (oracle/_ref, host mode = its own AVX path, nothing of this repo in the arithmetic): the full SPR search of
    tests/golden/search_expected.json with and without the option makes the same calls and ends on the same doubles and the same
    tree.  So the drop-in mirrors the gate (phl_lk.c, glue driver) and needs no device work for it."""
    import json, os, re, shutil, subprocess
    from conftest import GOLDEN, ROOT
    glue = os.path.join(ROOT, "oracle", "_ref", "phyml_glue_driver")
    if not os.path.exists(glue):
        pytest.skip("oracle/_ref not built (make -C oracle ref in the build container)")
    e = json.load(open(os.path.join(GOLDEN, "search_expected.json")))["search_nucleic_spr"]
    out = []
    for extra in ([], ["--alias_subpatt"]):
        wd = os.path.join(str(tmp_path), "a%d" % len(extra))
        os.makedirs(wd)
        shutil.copy(os.path.join(GOLDEN, "examples_nucleic.phy"), os.path.join(wd, "nucleic"))
        r = subprocess.run([glue] + e["driver_opts"] + ["--", "-i", "nucleic"] + e["phyml_args"] + extra, cwd=wd,
                           env=dict(os.environ, GLUE_MODE="host"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        m = re.search(r"GLUE_DRIVER (\{.*\})", r.stdout)
        assert r.returncode == 0 and m, r.stdout[-2000:]
        out.append(json.loads(m.group(1)))
    a, b = out
    assert a["calls"] == b["calls"] and a["calls"]["Update_Partial_Lk"] > 100000
    assert a["lnL_init"] == b["lnL_init"] and a["lnL_final"] == b["lnL_final"] and a["tree"] == b["tree"]


@pytest.mark.skipif(not os.path.exists("/root/reference/src/m4.c") or not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libphyml_ref.a")),
                    reason="build container only: needs the reference's sources and oracle/_ref")
def test_no_reference_behaviour_for_other_state_counts():
    """SURVEY 8 f4, second half: the only code of the reference that sets a state count other than 4 / 20 is the covarion model
    (mod->ns = n_o * n_h, src/init.c:6407).  oracle/probe_m4.sh tries its four doors -- the -DM4 build (src/cl.c does not compile),
    src/main.c:151-153's call order, src/interface.c:110-118's allocation, the same with the state count restored for Init_Model --
    and none reaches a likelihood function; the control (the generic loop on 4 states, which IS pinned) runs.  Should a later
    reference make one of them run, this test fails and the row reopens."""
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    out = subprocess.run(["bash", os.path.join(root, "oracle", "probe_m4.sh")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600).stdout
    assert "gcc exit status 1" in out and "opt_cov_alpha" in out, out
    doors = [l for l in out.splitlines() if l.startswith("exit status")]
    assert len(doors) == 3, out
    for l in doors:
        assert "exit status 139" in l and "likelihood functions in the backtrace: 0" in l and "REF_BENCH lines: 0" in l, out
    assert 'REF_BENCH {"lnL": -5680.0376' in out, out
